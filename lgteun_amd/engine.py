"""Host-side engine: flat parameter storage, plans, workspaces and the calls into the HIP library.

Layout decisions (MI355X-first):
  * all parameters of the net live in ONE flat fp32 device buffer in canonical state_dict order
    (each tensor 16-byte aligned); the nn.Parameters are views into it, so `state_dict()` /
    `load_state_dict()` / `.parameters()` keep working while kernels get base pointer + offsets,
    Adam is a single launch over two contiguous live ranges and the DDP bucket is the same two ranges.
  * gradients use an identical flat buffer; only live tensors (shared data module, eta, last stage's LGT)
    are ever written -- dead-stage parameters keep `grad is None` exactly like the reference.
  * one workspace tensor per (B, train) holds every activation; the library allocates nothing.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import (LG_FLAG_BWD_DATA, LG_FLAG_BWD_LGT, LG_FLAG_CHAINED, LG_FLAG_DEFER_DEAD, LG_FLAG_DROPOUT, LG_FLAG_FAITHFUL, LG_FLAG_SAVE,
                   LgConfig, check, variant_from_env)


def _block_names(pre):
    m = pre + '0.fn.'
    f = pre + '1.fn.'
    return [m + 'fn.local_mixer.pos_emb', m + 'fn.local_mixer.to_qkv.weight', m + 'fn.local_mixer.to_qkv.bias',
            m + 'fn.global_mixer.conv_amp.0.weight', m + 'fn.global_mixer.conv_amp.0.bias',
            m + 'fn.global_mixer.conv_pha.0.weight', m + 'fn.global_mixer.conv_pha.0.bias',
            m + 'fn.proj.weight', m + 'fn.proj.bias', m + 'norm.weight', m + 'norm.bias',
            f + 'fn.net.0.weight', f + 'fn.net.0.bias', f + 'fn.net.2.point_conv.weight', f + 'fn.net.2.point_conv.bias',
            f + 'fn.net.2.depth_conv.weight', f + 'fn.net.2.depth_conv.bias', f + 'fn.net.4.weight', f + 'fn.net.4.bias',
            f + 'norm.weight', f + 'norm.bias']


def canonical_names(C, K):
    """Pansharpening.state_dict() key order of the reference (models/unlg_former.py:22-48): 12 shared tensors,
    K eta, then 119 per stage.  This order defines the offsets table handed to lg_plan_create."""
    names = []
    for n in ('D.1', 'D.3', 'DT.1', 'DT.3'):
        names += [n + '.weight', n + '.bias']
    names += ['R.weight', 'R.bias', 'RT.weight', 'RT.bias']
    names += [f'eta.{i}' for i in range(K)]
    for i in range(K):
        p = f'prior_module.{i}.'
        names += [p + 'patch_embed.proj.0.weight', p + 'patch_embed.proj.0.bias', p + 'patch_embed.proj.1.weight',
                  p + 'patch_embed.proj.1.bias', p + 'patch_embed.norm.weight', p + 'patch_embed.norm.bias']
        names += _block_names(p + 'encoder_layers.0.0.blocks.0.')
        names += _block_names(p + 'encoder_layers.0.0.blocks.1.')
        names += [p + 'encoder_layers.0.1.1.weight', p + 'encoder_layers.0.1.1.bias']
        names += _block_names(p + 'bottleneck.blocks.0.')
        names += [p + 'decoder_layers.0.0.1.weight', p + 'decoder_layers.0.0.1.bias',
                  p + 'decoder_layers.0.1.weight', p + 'decoder_layers.0.1.bias']
        names += _block_names(p + 'decoder_layers.0.2.blocks.0.')
        names += _block_names(p + 'decoder_layers.0.2.blocks.1.')
        names += [p + 'tail.1.weight', p + 'tail.1.bias']
    return names


def flat_layout(names, numels, K):
    """offsets (floats, each tensor 16-byte aligned) of the canonical tensors in the flat buffers, total size, indices of
    the live tensors and the two contiguous live ranges [(shared + eta), (last stage's LGT)]  (SURVEY D3)."""
    offs, total = [], 0
    for n in numels:
        offs.append(total)
        total += (max(n, 1) + 3) // 4 * 4
    n_head = 12 + K
    first_last = 12 + K + 119 * (K - 1)
    live_idx = list(range(n_head)) + list(range(first_last, len(names)))
    live_ranges = [(0, offs[n_head] if n_head < len(offs) else total), (offs[first_last], total)]
    return offs, total, live_idx, live_ranges


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class Engine:
    def __init__(self, module):
        self.lib = _lib.lib()          # raises if the HIP library is not built -- no fallback
        self.C = int(module.in_channels)
        self.K = int(module.stage)
        self.module_mode = lambda: module.mode
        self.module_faithful_eval = lambda: getattr(module, 'faithful_eval', False)
        self.module_precision = lambda: getattr(module, 'precision', 'fp32')
        names = canonical_names(self.C, self.K)
        params = dict(module.named_parameters())
        if set(names) != set(params):
            raise RuntimeError('parameter surface does not match the canonical LGTEUN state_dict')
        dev = params[names[0]].device
        if dev.type != 'cuda':
            raise RuntimeError('lgteun_amd: parameters must live on an MI355X (cuda) device')
        for n in names:
            if params[n].device != dev or params[n].dtype != torch.float32:
                raise RuntimeError(f'parameter {n}: expected float32 on {dev}')
        self.device = dev
        self.names = names
        offs, total, d3_idx, d3_ranges = flat_layout(names, [params[n].numel() for n in names], self.K)
        self.offsets = offs
        self.total = total
        # which tensors get a gradient: the reference's graph (shared + eta + last LGT) or, in 'chained' mode, all
        self._live = {False: (d3_idx, d3_ranges), True: (list(range(len(names))), [(0, total)])}
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params = [params[n] for n in names]
        for n, o in zip(names, offs):
            p = params[n]
            view = self.flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        # gradients + (one 16-byte slot behind them) the loss scalar of the fused step: ONE buffer, so that train_step clears both with one fill launch
        self._gbuf = torch.zeros(total + 4, dtype=torch.float32, device=dev)
        self.gflat = self._gbuf[:total]
        self._ranges_dev = {ch: torch.tensor([v for r in rg for v in r], dtype=torch.int64, device=dev)
                            for ch, (_, rg) in self._live.items()}
        self._plans = {}
        self._ws = {}
        self._ws_pool = {}             # autograd path: released training workspaces by (plan, B, train), see _WsLease
        self._loss = self._gbuf[total:total + 1]
        self._seed_ctr = 0
        # opt-in (LG_OVERLAP_DEAD=1 / engine.overlap_dead = True): 'faithful' training enqueues the K-1 dead-stage LGT forwards on a
        # second stream behind the LGT backward, beside the K data-step backwards + Adam (a chain of small latency-bound launches).
        # Bitwise the same step (tested).  Measured +0.9 % pairs/s only -- the persistent forward kernels fill every CU's LDS, so the
        # small launches get in at kernel boundaries -- while the co-running launches stretch the fused FFN's measured duration by
        # 6 %, so the default keeps one stream and per-kernel numbers that mean what they say.
        self.overlap_dead = os.environ.get('LG_OVERLAP_DEAD', '0') == '1'
        self.variant = None            # lg_config.variant of the plans: None = from the diagnostic LG_* environment variables (normally 0)
        self._side_stream = None
        self.world = 1
        self.rank = 0
        self.process_group = None
        self.buckets = None
        self.force_collectives = False   # attach_ddp(force=True): collectives also in a group of one rank
        self.local_only = False        # True: the caller runs independent replicas inside an initialised process group on purpose

    def chained(self):
        return self.module_mode() == 'chained'

    @property
    def live_idx(self):
        return self._live[self.chained()][0]

    @property
    def live_ranges(self):
        return self._live[self.chained()][1]

    @property
    def ranges_dev(self):
        return self._ranges_dev[self.chained()]

    @property
    def max_range(self):
        return max(b - a for a, b in self.live_ranges)

    def attach_ddp(self, group=None, broadcast=True, force=False):
        """join a torch.distributed group: reduce the flat gradient buffer every step.  broadcast=True (an explicit
        `module.attach_ddp()`): rank-0 weights go to every rank -- a COLLECTIVE, so every rank of the group must make the call.
        broadcast=False (an engine rebuilt under an attached module, e.g. after `.to()`): no communication; the ranks' weights
        were made equal by the first attachment and identical updates keep them equal.
        force=True: issue the collectives in a group of ONE rank too (the broadcast here, the all-reduce(s) of every train step) --
        how a single-GPU box exercises the RCCL communicator and the bucket code an 8-GPU run takes (tests, `LGTEUN_FORCE_PG`)."""
        import torch.distributed as dist
        from .ddp import GradBuckets, broadcast_flat
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.process_group = group
        self.force_collectives = bool(force and dist.is_initialized())
        if self.world > 1 or self.force_collectives:
            if broadcast:
                broadcast_flat(self.flat, 0, group, force=self.force_collectives)
            self.buckets = {ch: GradBuckets(rg, group) for ch, (_, rg) in self._live.items()}
        return self

    def global_loss(self):
        """the GLOBAL-mean loss of the last train_step as a Python float (host sync; under DDP one scalar all-reduce: `_loss`
        holds this rank's share of the global mean).  For the logging cadence only."""
        t = self._loss.clone()
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
        return float(t.item())

    def _check_attached(self):
        """a process group with more than one rank exists but this engine never joined it: every rank would train on its own
        shard without the gradient all-reduce and silently diverge (reference: nn.DataParallel reduces implicitly,
        base_model.py:91-100)"""
        if self.world > 1 or self.local_only:
            return
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise RuntimeError('torch.distributed is initialised with world size > 1 but this engine is not attached: call '
                               'module.attach_ddp() (or set engine.local_only = True for independent replicas)')

    # ------------------------------------------------------------------------------------------
    def valid(self):
        """every parameter is still the view into the flat buffer it was given (a caller that re-assigns one tensor's `.data`,
        `.to()`, a re-created parameter: the kernels would read stale weights) -- 492 pointer compares at K = 4, ~50 us"""
        flat_ptr = self.flat.data_ptr()
        return all(p.data_ptr() == flat_ptr + 4 * o for p, o in zip(self.params, self.offsets))

    def __del__(self):
        try:
            for pl in self._plans.values():
                self.lib.lg_plan_destroy(pl)
        except Exception:  # noqa: BLE001
            pass

    def plan(self, H, W):
        prec = {'fp32': 0, 'bf16': 1}[self.module_precision()]
        var = variant_from_env() if self.variant is None else int(self.variant)
        key = (H, W, prec, var)
        if key not in self._plans:
            cfg = LgConfig(self.C, self.K, H, W, prec, var)
            arr = (ctypes.c_int64 * len(self.offsets))(*self.offsets)
            out = ctypes.c_void_p()
            check(self.lib.lg_plan_create(ctypes.byref(cfg), arr, len(self.offsets), ctypes.byref(out)), 'lg_plan_create')
            self._plans[key] = out
        return self._plans[key]

    def workspace(self, plan, B, train):
        """train: 0 inference, 1 training, 2 chained training (K saved activation sets)"""
        need = self.lib.lg_workspace_bytes(plan, B, int(train))
        key = (plan.value, B, int(train))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def lease_workspace(self, plan, B, train):
        """a training workspace of its own for one autograd graph: the saved activations of a forward must survive until ITS
        backward, whatever other forwards (a second graph with the same B, Engine.train_step) run in between.  The lease
        returns the buffer to a pool when the autograd context that holds it is freed."""
        need = self.lib.lg_workspace_bytes(plan, B, int(train))
        key = (plan.value, B, int(train))
        pool = self._ws_pool.setdefault(key, [])
        ws = pool.pop() if pool else None
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return _WsLease(ws, pool)

    def next_seed(self):
        """dropout counter seed: torch's seed, a per-step counter and the DDP rank (ranks seeded alike must not draw the same
        masks for their shards)"""
        self._seed_ctr += 1
        return (torch.initial_seed() * 0x9E3779B1 + self._seed_ctr + self.rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def _check_inputs(self, ms, pan):
        if ms.dim() != 4 or pan.dim() != 4 or ms.shape[1] != self.C or pan.shape[1] != 1:
            raise ValueError(f'expected ms [B,{self.C},h,w] and pan [B,1,4h,4w], got {tuple(ms.shape)} / {tuple(pan.shape)}')
        B, _, h, w = ms.shape
        if pan.shape[0] != B or pan.shape[2] != 4 * h or pan.shape[3] != 4 * w:
            raise ValueError('pan must be 4x the MS size')
        if ms.dtype != torch.float32 or pan.dtype != torch.float32:
            raise ValueError('inputs must be float32 (NCHW), like the reference')
        if ms.device != self.device or pan.device != self.device:
            raise ValueError(f'inputs must be on {self.device}')
        return B, 4 * h, 4 * w

    # ------------------------------------------------------------------------------------------
    def forward_raw(self, ms, pan, flags, seed=0, lease=False):
        B, H, W = self._check_inputs(ms, pan)
        ms = ms.contiguous()
        pan = pan.contiguous()
        plan = self.plan(H, W)
        train = (2 if flags & LG_FLAG_CHAINED else 1) if flags & LG_FLAG_SAVE else 0
        if lease:
            holder = self.lease_workspace(plan, B, train)
            ws = holder.ws
        else:
            holder = None
            ws = self.workspace(plan, B, train)
        out = torch.empty(B, self.C, H, W, dtype=torch.float32, device=self.device)
        check(self.lib.lgteun_forward(plan, _ptr(self.flat), _ptr(ms), _ptr(pan), _ptr(out), _ptr(ws), ws.numel(), B, flags,
                                      seed, _stream_ptr()), 'lgteun_forward')
        return out, (plan, ws, ms, pan, B, holder)

    def backward_raw(self, saved, dout, gflat, flags, seed=0):
        plan, ws, ms, pan, B = saved[:5]
        dout = dout.contiguous()
        check(self.lib.lgteun_backward(plan, _ptr(self.flat), _ptr(gflat), _ptr(ms), _ptr(pan), _ptr(dout), _ptr(ws),
                                       ws.numel(), B, flags, seed, _stream_ptr()), 'lgteun_backward')

    def base_flags(self, training):
        mode = self.module_mode()
        if mode not in ('faithful', 'live', 'chained'):
            raise ValueError(f"mode must be 'faithful', 'live' or 'chained' (got {mode!r})")
        f = {'faithful': LG_FLAG_FAITHFUL, 'live': 0, 'chained': LG_FLAG_CHAINED}[mode]
        if training:
            f |= LG_FLAG_DROPOUT
        return f

    def forward_autograd(self, ms, pan, training):
        flags = self.base_flags(training)
        need_grad = torch.is_grad_enabled() and any(self.params[i].requires_grad for i in self.live_idx)
        if not need_grad:
            if not training and not self.module_faithful_eval():
                # inference: the K-1 dead-stage LGTs change nothing in the output (bitwise, tested) -- they run only
                # where the reference's WORK is being reproduced (training in 'faithful' mode, or module.faithful_eval = True)
                flags &= ~LG_FLAG_FAITHFUL
            out, _ = self.forward_raw(ms, pan, flags, self.next_seed() if training else 0)
            return out
        live = [self.params[i] for i in self.live_idx]
        return _LgteunFn.apply(self, ms, pan, flags, *live)

    # ------------------------------------------------------------------------------------------
    def train_step(self, ms, pan, gt, optim, loss_weight=1.0):
        """forward + L1(mean) + backward + Adam as library calls; returns the device loss scalar (this rank's share)."""
        self._check_attached()
        flags = self.base_flags(True) | LG_FLAG_SAVE
        if not getattr(optim, 'dropout', True):
            flags &= ~LG_FLAG_DROPOUT
        seed = self.next_seed()
        self._gbuf.zero_()                 # gradients and the loss scalar
        defer = bool(self.overlap_dead and (flags & LG_FLAG_FAITHFUL) and not (flags & LG_FLAG_CHAINED) and self.K > 1)
        if defer:
            flags |= LG_FLAG_DEFER_DEAD
        out, saved = self.forward_raw(ms, pan, flags, seed)
        gt = gt.contiguous()
        dout = torch.empty_like(out)
        n_local = out.numel()
        check(self.lib.lg_l1_loss(_ptr(out), _ptr(gt), _ptr(dout), _ptr(self._loss), n_local, n_local * self.world,
                                  float(loss_weight), _stream_ptr()), 'lg_l1_loss')
        bk = self.buckets[bool(flags & LG_FLAG_CHAINED)] if (self.world > 1 or self.force_collectives) else None
        overlap = bool(bk is not None and bk.overlap and not (flags & LG_FLAG_CHAINED))
        if defer or overlap:
            # two backward calls: the dead-stage forwards (side stream) and / or the opt-in asynchronous bucket of the last stage's
            # LGT start behind the LGT backward and run beside the K data-step backwards
            self.backward_raw(saved, dout, self.gflat, flags | LG_FLAG_BWD_LGT, seed)
            if defer:
                self._dead_forward(saved, flags, seed)
            if overlap:
                bk.start(self.gflat, 1)
            self.backward_raw(saved, dout, self.gflat, flags | LG_FLAG_BWD_DATA, seed)
            if overlap:
                if bk.serial:
                    bk.finish()            # the LGT bucket's result is in before the shared bucket starts
                bk.start(self.gflat, 0)
                bk.finish()
        else:
            self.backward_raw(saved, dout, self.gflat, flags, seed)
        if bk is not None and not overlap:
            bk.all_reduce(self.gflat)      # default: ONE stream-ordered collective behind the whole backward (ddp.py)
        optim.step_flat(self)
        if defer:
            torch.cuda.current_stream().wait_stream(self._side_stream)   # the step ends when its dead-stage work has ended
        return self._loss

    def _dead_forward(self, saved, flags, seed):
        """the dead-stage LGT forwards of a LG_FLAG_DEFER_DEAD forward, on the side stream: ordered behind everything enqueued so far
        (the LGT backward has read the saved activations they overwrite), concurrent with what the caller enqueues next"""
        plan, ws, _, _, B = saved[:5]
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            check(self.lib.lgteun_dead_forward(plan, _ptr(self.flat), _ptr(ws), ws.numel(), B, flags, seed, _stream_ptr()),
                  'lgteun_dead_forward')

    def adam(self, state, step, lr, betas, eps, grad_scale=1.0):
        check(self.lib.lg_adam_step(_ptr(self.flat), _ptr(self.gflat), _ptr(state['exp_avg']), _ptr(state['exp_avg_sq']),
                                    _ptr(self.ranges_dev), len(self.live_ranges), self.max_range, step, float(lr),
                                    float(betas[0]), float(betas[1]), float(eps), float(grad_scale), _stream_ptr()),
              'lg_adam_step')


class _WsLease:
    """holds one training workspace for the lifetime of an autograd graph; gives it back to the engine's pool afterwards"""

    def __init__(self, ws, pool):
        self.ws, self._pool = ws, pool

    def __del__(self):
        try:
            if not self._pool:             # one spare buffer per (plan, B, train); extras go back to the allocator
                self._pool.append(self.ws)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class _LgteunFn(torch.autograd.Function):
    """autograd bridge for callers that use torch optimizers / losses on the module output."""

    @staticmethod
    def forward(ctx, engine, ms, pan, flags, *live):
        engine._check_attached()
        seed = engine.next_seed() if (flags & LG_FLAG_DROPOUT) else 0
        out, saved = engine.forward_raw(ms, pan, flags | LG_FLAG_SAVE, seed, lease=True)   # this graph's own activations
        ctx.engine, ctx.saved, ctx.flags, ctx.seed = engine, saved, flags | LG_FLAG_SAVE, seed
        ctx.live_idx = list(engine.live_idx)    # the mode may change before backward runs
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.engine
        g = torch.zeros(eng.total, dtype=torch.float32, device=eng.device)
        eng.backward_raw(ctx.saved, dout, g, ctx.flags, ctx.seed)
        if eng.world > 1:
            # the caller's loss is this rank's local mean: average over ranks like torch DDP does
            eng.buckets[bool(ctx.flags & LG_FLAG_CHAINED)].all_reduce(g)
            g.div_(eng.world)
        grads = []
        for i in ctx.live_idx:
            p = eng.params[i]
            o = eng.offsets[i]
            grads.append(g[o:o + p.numel()].view(p.shape))
        return (None, None, None, None, *grads)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (reference models/base/base_model.py:123-124) as one HIP launch over the flat
    live ranges.  Subclasses Optimizer so lr_scheduler.StepLR (base_model.py:137-147) drives `param_groups[0]['lr']`."""
    is_fused_lgteun = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps))
        self._step = 0
        self._state = None
        self.dropout = True

    def step_flat(self, engine):
        if self._state is None or self._state['exp_avg'].numel() != engine.total:
            self._state = dict(exp_avg=torch.zeros_like(engine.flat), exp_avg_sq=torch.zeros_like(engine.flat))
        elif self._state['exp_avg'].device != engine.flat.device or not self._state['exp_avg'].is_contiguous():
            # moments restored from a checkpoint (loaded to the host): the kernel takes device pointers
            self._state = {k: v.to(engine.flat.device).contiguous() for k, v in self._state.items()}
        self._step += 1
        g = self.param_groups[0]
        engine.adam(self._state, self._step, g['lr'], g['betas'], g['eps'])

    def step(self, closure=None):   # pragma: no cover - the fused path goes through Engine.train_step
        raise RuntimeError('FusedAdam is stepped by Engine.train_step(); use torch.optim.Adam for the autograd path')

    def state_dict(self):
        sd = super().state_dict()
        sd['lgteun'] = dict(step=self._step, state=self._state)
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)                   # the caller's dict stays as it was
        extra = sd.pop('lgteun', None)
        super().load_state_dict(sd)
        if extra is not None:
            self._step, self._state = extra['step'], extra['state']
