"""Runner counterpart of reference models/base/base_model.py restricted to the hot path's callers:
add_module :56, set_cuda :91 (one process per GPU + RCCL instead of nn.DataParallel), load_checkpoint :102,
set_optim :116 (Adam -> fused HIP Adam), set_sched :137 (StepLR stepped EVERY iteration :197-199),
train :164, test :267 (reference-based indices PSNR / SSIM / Q / SAM / ERGAS on the reduced-resolution set, the no-reference
D_lambda / D_s / QNR on the full-resolution set), save :354 (same `train_out/` location; optimizer state added)."""
import os.path as osp

import numpy as np
import torch
import torch.nn as nn
from torch.optim import SGD, Adam, AdamW, RMSprop, lr_scheduler

from . import metrics as mtc
from .compat import Timer, mkdir_or_exist
from .losses import get_loss_module


def data_normalize(img_dict, bit_depth):
    """reference dataset/utils.py:232-249"""
    max_value = 2 ** bit_depth - .5
    return {k: (v if k == 'image_id' else v / max_value) for k, v in img_dict.items()}


def data_denormalize(img, bit_depth):
    """reference dataset/utils.py:252-263"""
    return img * (2 ** bit_depth - .5)


def smart_time(second):
    second = int(second)
    return f'{second // 3600}h {second % 3600 // 60}m {second % 60}s'


class Base_model:
    """The attribute names are the reference's (subclasses and configs rely on them: base_model.py:26-54); how they are filled is this
    build's."""
    OUT_DIRS = ('train_out', 'test_out0', 'test_out1')            # under <work_dir>/<datas>/ (base_model.py:44-46)
    FREQ_DEFAULTS = dict(save_freq=10000, test_freq=10000, eval_freq=10000, max_iter=100000)

    def __init__(self, cfg, logger, train_data_loader, test_data_loader0, test_data_loader1):
        self.cfg, self.logger = cfg, logger
        self.work_dir, self.datas = cfg.work_dir, cfg.datas
        self.train_data_loader, self.test_data_loader0, self.test_data_loader1 = train_data_loader, test_data_loader0, test_data_loader1
        mkdir_or_exist(self.work_dir)
        for d in self.OUT_DIRS:
            setattr(self, d, f'{self.work_dir}/{self.datas}/{d}')
        self.module_dict, self.optim_dict, self.sched_dict, self.switch_dict, self.eval_results = {}, {}, {}, {}, {}
        self.loss_module = get_loss_module(full_cfg=cfg, logger=logger)
        self.last_iter = 0
        self.rank, self.world = 0, 1      # one process per GPU: set_cuda() reads them from torch.distributed
        self.timer = Timer()

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def add_module(self, module_name, module, switch=True):
        assert isinstance(module, nn.Module)
        self.module_dict[module_name] = module
        self.switch_dict[module_name] = switch

    def print_total_params(self):
        count = sum(sum(p.numel() for p in m.parameters()) for m in self.module_dict.values())
        self.logger.info(f'total params: {count},{round(count / (1000 ** 2), 4)} M')
        return count

    def init(self):
        pass

    def set_cuda(self):
        """one process per GPU (torch.distributed/RCCL set up by the launcher) -- replaces nn.DataParallel."""
        dev = torch.device('cuda', torch.cuda.current_device())
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if ddp else (0, 1)
        for name in self.module_dict:
            self.module_dict[name] = self.module_dict[name].to(dev)
            if ddp and hasattr(self.module_dict[name], 'attach_ddp'):
                self.module_dict[name].attach_ddp()       # rank-0 weights broadcast; gradients all-reduced every step
        for name in self.loss_module:
            self.loss_module[name] = self.loss_module[name].to(dev)

    def _load_modules(self, checkpoint):
        for module_name, module in self.module_dict.items():
            src = checkpoint[module_name]
            module.load_state_dict(src.state_dict() if hasattr(src, 'state_dict') else src)

    def _read_checkpoint(self, path, allow_pickle):
        """plain-tensor checkpoints (what save() writes; what tools/convert_checkpoint.py makes of a reference-era file) load with
        `weights_only=True`: nothing in the file is executed.  The reference's own format -- whole pickled module objects,
        base_model.py:362-368 -- runs code from the file while loading, so it is read only when the caller says the file is
        trusted: `allow_pickle=True` or `cfg.allow_pickled_checkpoint = True`."""
        if allow_pickle is None:
            allow_pickle = bool(self.cfg.get('allow_pickled_checkpoint', False))
        import pickle
        try:
            return torch.load(path, map_location='cpu', weights_only=True)
        except (pickle.UnpicklingError, RuntimeError) as e:   # what torch raises for anything beyond tensors / containers; a missing or
            if isinstance(e, RuntimeError) and 'weights_only' not in str(e).lower() and 'unsupported' not in str(e).lower() \
                    and 'global' not in str(e).lower():        # unreadable file (OSError, EOFError, a corrupt archive) is NOT a reason to unpickle
                raise
            if not allow_pickle:
                raise RuntimeError(
                    f'{path} is not a plain-tensor checkpoint (the reference pickles whole module objects). Convert it once with '
                    'tools/convert_checkpoint.py, or pass allow_pickle=True / set cfg.allow_pickled_checkpoint for a file you '
                    f'trust.  ({type(e).__name__}: {str(e)[:200]})') from e
        if self.logger is not None:
            self.logger.warning(f'{path}: not a plain-tensor checkpoint, unpickling it as allowed by allow_pickle / cfg.allow_pickled_checkpoint')
        return torch.load(path, map_location='cpu', weights_only=False)

    def load_checkpoint(self, path, allow_pickle=None):
        checkpoint = self._read_checkpoint(path, allow_pickle)
        self.last_iter = checkpoint['iter_num']
        self._load_modules(checkpoint)
        self._resume_optim = checkpoint.get('optim')     # checkpoints of this build carry it; restored by set_optim()
        for name, st in (self._resume_optim or {}).items():
            if name in self.optim_dict:
                self.optim_dict[name].load_state_dict(st)

    def load_pretrained(self, path, allow_pickle=None):
        self._load_modules(self._read_checkpoint(path, allow_pickle))

    def set_optim(self):
        from .engine import FusedAdam
        optim_cfg = self.cfg.get('optim_cfg', {})
        for module_name, module in self.module_dict.items():
            if module_name in optim_cfg:
                cfg = dict(optim_cfg[module_name])
                typ = cfg.pop('type')
                fused = cfg.pop('fused', True)
                if typ == 'Adam':
                    self.optim_dict[module_name] = (FusedAdam if fused else Adam)(module.parameters(), **cfg)
                elif typ == 'RMSprop':
                    self.optim_dict[module_name] = RMSprop(module.parameters(), **cfg)
                elif typ == 'SGD':
                    self.optim_dict[module_name] = SGD(module.parameters(), **cfg)
                elif typ == 'AdamW':
                    self.optim_dict[module_name] = AdamW(module.parameters(), **cfg)
                else:
                    raise SystemExit(f'No such type optim:{typ}')
            else:
                self.optim_dict[module_name] = Adam(module.parameters(), betas=(0.9, 0.999), lr=1e-4)
            resume = getattr(self, '_resume_optim', None) or {}
            if module_name in resume:                     # load_checkpoint ran first (main.py order): continue the Adam moments
                self.optim_dict[module_name].load_state_dict(resume[module_name])

    def set_sched(self):
        sched_cfg = dict(self.cfg.get('sched_cfg', dict(step_size=10000, gamma=0.99)))
        for name, optim in self.optim_dict.items():
            self.sched_dict[name] = lr_scheduler.StepLR(optimizer=optim, **sched_cfg)

    def _device(self):
        return next(next(iter(self.module_dict.values())).parameters()).device

    def _due(self, freq, iter_id):
        """a periodic action of the training loop is due at this iteration (never at the last one, -1 switches it off: base_model.py:187-195)"""
        return freq != -1 and iter_id % freq == 0 and iter_id != self.cfg.max_iter

    def _train_batches(self, dev):
        """(iteration number, normalised device batch) from last_iter + 1 up to cfg.max_iter, cycling through the training loader"""
        it = self.last_iter
        while it < self.cfg.max_iter:
            for batch in self.train_data_loader:
                it += 1
                batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                yield it, data_normalize(batch, self.cfg.bit_depth)           # normalised unconditionally (base_model.py:181)
                if it >= self.cfg.max_iter:
                    return

    def train(self):
        """the training cadence of the reference (base_model.py:164-204): train_iter on every batch, then save / evaluate when due (the
        evaluation on the full-resolution set first, then the reduced-resolution one; fused images are written when test_freq is due too),
        then one StepLR tick per ITERATION for every switched-on module"""
        for key, default in self.FREQ_DEFAULTS.items():
            self.cfg.setdefault(key, default)
        self.timer = Timer()
        for iter_id, input_batch in self._train_batches(self._device()):
            for module in self.module_dict.values():
                module.train()
            self.train_iter(iter_id=iter_id, input_batch=input_batch)
            if self._due(self.cfg.save_freq, iter_id):
                self.save(iter_id=iter_id)
            if self._due(self.cfg.eval_freq, iter_id):
                write = self._due(self.cfg.test_freq, iter_id)
                for ref in (False, True):
                    self.test(iter_id=iter_id, save=write, ref=ref)
            for name, sched in self.sched_dict.items():
                if self.switch_dict[name]:
                    sched.step()

    def print_train_log(self, iter_id, loss_res, log_freq=10):
        if iter_id % log_freq == 0 and self.logger is not None and self.rank == 0:
            avg_iter_time = self.timer.since_last_check() / log_freq
            remain_time = avg_iter_time * (self.cfg.max_iter - iter_id)
            lr = self.optim_dict['core_module'].param_groups[0]['lr']
            self.logger.info(f'iteration {iter_id} of {self.cfg.max_iter} | lr {lr:.6f} | full loss {loss_res["full_loss"]:.6f} | '
                             f'time left {smart_time(remain_time)}')

    def get_model_output(self, input_batch):
        raise NotImplementedError

    def train_iter(self, iter_id, input_batch, log_freq=10):
        raise NotImplementedError

    @torch.no_grad()
    def test(self, iter_id, save=False, ref=True):
        """evaluation (base_model.py:267-352).  ref=True: reduced-resolution set, PSNR / SSIM / Q / SAM / ERGAS against the target,
        `<metric>_mean` / `<metric>_std` lists in self.eval_results like the reference.  ref=False: the full-resolution set, no target:
        D_lambda / D_s / QNR from the fused image, the PAN and the MS input (lgteun_amd/metrics.py: parity-unpinned).  Inputs are always normalised; arrays are brought back
        to digital numbers before the metrics / the TIFF writer only with cfg.norm_input (base_model.py:296,311-316)."""
        from .dataset import save_image
        loader = self.test_data_loader1 if ref else self.test_data_loader0
        for module in self.module_dict.values():
            module.eval()
        dev = self._device()
        names = ['PSNR', 'SSIM', 'Q', 'SAM', 'ERGAS'] if ref else ['D_lambda', 'D_s', 'QNR']
        denorm = bool(self.cfg.get('norm_input', False))
        out_dir = osp.join(self.test_out1 if ref else self.test_out0, f'iter_{iter_id}')
        # one process per GPU: the evaluation set is SPLIT over the ranks (the reference is one process).  A loader built with this rank's
        # ShardedSampler (dataset.build_loader(rank, world)) or declared so with cfg.eval_sharded already yields this rank's share; of any
        # other loader (the same on every rank) rank r takes batches r, r + world, ...  Every rank writes the fused images of ITS share,
        # the per-image metric rows of all ranks are gathered in rank order before the mean / std.
        if save:
            mkdir_or_exist(out_dir)
        from .dataset import ShardedSampler
        inner = getattr(loader, 'loader', loader)
        sharded = self.world > 1 and (bool(self.cfg.get('eval_sharded', False)) or isinstance(getattr(inner, 'sampler', None), ShardedSampler))

        def to_np(t):   # [b c h w] -> [b h w c]
            t = data_denormalize(t, self.cfg.bit_depth) if denorm else t
            return t.permute(0, 2, 3, 1).cpu().numpy()
        # what identifies an image ACROSS ranks (`image_id` is a file-name prefix -- two directories may hold equal ones -- and may be
        # absent): the dataset index this rank's ShardedSampler hands out, in the order the loader consumes it; for a loader every rank iterates
        # in full (batches dealt out round-robin) the position in that common sequence; for a foreign pre-sharded loader nothing is known,
        # its rows are kept as they come (rank-tagged keys never collide)
        smp = getattr(inner, 'sampler', None)
        own_idx = [int(i) for i in smp] if (sharded and isinstance(smp, ShardedSampler)) else None
        res, ids, seen_here = [], [], 0
        for bi, input_batch in enumerate(loader or []):
            if self.world > 1 and not sharded and bi % self.world != self.rank:
                continue
            input_batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in input_batch.items()}
            input_batch = data_normalize(input_batch, self.cfg.bit_depth)
            out = to_np(self.get_model_output(input_batch))
            if ref:
                gt = to_np(input_batch['target'])
                res.extend(mtc.ref_evaluate(out[i], gt[i]) for i in range(out.shape[0]))
            else:                                                # full-resolution set: no target (base_model.py:330-334)
                pan_np, lr_np = to_np(input_batch['input_pan']), to_np(input_batch['input_lr'])
                res.extend(mtc.no_ref_evaluate(out[i], pan_np[i], lr_np[i]) for i in range(out.shape[0]))
            nb = out.shape[0]
            if own_idx is not None:
                ids.extend(('idx', own_idx[seen_here + i]) for i in range(nb))
            elif not sharded:
                ids.extend(('pos', bi, i) for i in range(nb))
            else:
                ids.extend(('rank', self.rank, seen_here + i) for i in range(nb))
            seen_here += nb
            if save:
                for i, image_id in enumerate(input_batch['image_id']):
                    # [C, H, W] for the writer (the reference hands its HWC array to a CHW writer, base_model.py:336: a
                    # transposed file; not reproduced)
                    save_image(osp.join(out_dir, f'{image_id}_mul_hat.tif'), np.moveaxis(out[i], -1, 0))
        latest = {}
        if self.world > 1:
            import torch.distributed as dist
            rows = [None] * self.world
            dist.all_gather_object(rows, [(tuple(i), list(map(float, r))) for i, r in zip(ids, res)])
            # an image that reached two ranks (a PADDED sharded sampler wraps around when world does not divide the set: build the
            # evaluation loaders with build_loader(..., evaluation=True)) is scored once, like the reference's one-process loop:
            # told by its dataset index, never by its name
            seen, res = set(), []
            for part in rows:
                for i, r in part:
                    if i not in seen:
                        seen.add(i)
                        res.append(r)
        if res:
            res = np.array(res)
            for k, name in enumerate(names):
                self.eval_results.setdefault(f'{name}_mean', []).append(round(float(res[:, k].mean()), 4))
                self.eval_results.setdefault(f'{name}_std', []).append(round(float(res[:, k].std()), 4))
                latest[name] = (float(res[:, k].mean()), float(res[:, k].std()))
            if self.logger is not None and self.rank == 0:
                self.logger.info(f"iter {iter_id} {'low' if ref else 'full'}-resolution eval: {latest}")
        self._barrier()                    # no rank runs ahead of files another rank is still writing
        return latest

    def save(self, iter_id):
        """`train_out/model_iter_N.pth` like the reference (base_model.py:354-369), written by rank 0 only, holding plain tensors:
        {module name: state_dict, 'iter_num', 'optim': optimizer state (absent in the reference)} -- loadable with
        `weights_only=True`.  `cfg.pickle_modules = True` writes the reference's own format instead (whole pickled module
        objects), for tools that expect it; such a file needs `allow_pickle` to be read back."""
        path = osp.join(self.train_out, f'model_iter_{iter_id}.pth')
        if self.rank == 0:
            mkdir_or_exist(self.train_out)
            ckpt = {'iter_num': iter_id}
            pickled = bool(self.cfg.get('pickle_modules', False))
            for name, module in self.module_dict.items():
                core = module.module if hasattr(module, 'module') else module
                ckpt[name] = core if pickled else {k: v.detach().cpu() for k, v in core.state_dict().items()}
            ckpt['optim'] = {k: v.state_dict() for k, v in self.optim_dict.items()}
            tmp = path + '.tmp'
            torch.save(ckpt, tmp)
            import os
            os.replace(tmp, path)          # readers never see a half-written file
        self._barrier()
        return path
