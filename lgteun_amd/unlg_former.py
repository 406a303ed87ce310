"""Drop-in mirror of the reference's `models/unlg_former.py` surface for the MI355X-native path.

`Pansharpening` keeps the reference's constructor (cfg, logger, stage), call signature
`core_module(ms, pan)` (MS first, PAN second; unlg_former.py:84,94), NCHW fp32 I/O and -- key for
checkpoints (base_model.py:102-114,354-369) -- the identical `state_dict()` keys and shapes
(492 tensors at K=4).  The module tree below is built from stock torch layers used
ONLY as parameter containers (same default initialisation as the reference gets from PyTorch;
pos_emb ~ trunc_normal(0,1,[-2,2]) LGT.py:127-128, eta = 0.1 unlg_former.py:40).  No layer's
forward is ever called: compute goes through the HIP library (engine.py).  There is no CPU path.

`UnlgFormer` mirrors the runner wrapper (unlg_former.py:70-113): registered in MODELS under the
same name, same 5-argument constructor, `get_model_output` / `train_iter` with the same batch keys.
"""
import torch
import torch.nn as nn

from .base_model import Base_model
from .builder import MODELS
from .engine import Engine, canonical_names


def _point_conv(cin, cout):          # bmu.point_conv   basic_module_unformer_v2.py:13
    return nn.Conv2d(cin, cout, 1, 1, 0, groups=1)


def _dep_conv(c, k):                 # bmu.dep_conv     basic_module_unformer_v2.py:17
    return nn.Conv2d(c, c, k, 1, k // 2, groups=c)


class _Slot(nn.Identity):
    """Parameter-less placeholder keeping the reference's Sequential indices (sampling_unit_)."""


class _Box(nn.Module):
    """Named container (children are assigned as attributes)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            setattr(self, k, v)


class _LocalMixerParams(nn.Module):  # LGT.py:112-128
    def __init__(self, channels, win, heads):
        super().__init__()
        self.to_qkv = _point_conv(channels, channels * 3)
        self.pos_emb = nn.Parameter(torch.empty(1, heads, win * win, win * win))
        nn.init.trunc_normal_(self.pos_emb, mean=0.0, std=1.0, a=-2.0, b=2.0)


def _lgb(channels, nblocks, win, heads):     # LGT.py:222-248
    blocks = nn.ModuleList()
    for _ in range(nblocks):
        half = channels // 2
        mixer = _Box(
            local_mixer=_LocalMixerParams(half, win, heads),
            global_mixer=_Box(conv_amp=nn.Sequential(_dep_conv(half, 1)), conv_pha=nn.Sequential(_dep_conv(half, 1))),
            proj=_point_conv(channels, channels))
        ffn = _Box(net=nn.Sequential(
            _point_conv(channels, channels * 4), _Slot(),
            _Box(point_conv=_point_conv(channels * 4, channels * 4), depth_conv=_dep_conv(channels * 4, 3)),
            _Slot(), _point_conv(channels * 4, channels)))
        blocks.append(nn.ModuleList([
            _Box(fn=_Box(fn=mixer, norm=nn.LayerNorm(channels))),
            _Box(fn=_Box(fn=ffn, norm=nn.LayerNorm(channels)))]))
    return _Box(blocks=blocks)


def _lgt(in_channels, embed, win=8, num_block=(2, 1), heads=2):   # LGT.py:251-303
    m = _Box()
    m.patch_embed = _Box(proj=nn.Sequential(_dep_conv(in_channels, 1), _point_conv(in_channels, embed)),
                         norm=nn.LayerNorm(embed))
    m.encoder_layers = nn.ModuleList([nn.ModuleList([
        _lgb(embed, num_block[0], win, heads), nn.Sequential(_Slot(), _point_conv(embed, embed * 2))])])
    m.bottleneck = _lgb(embed * 2, num_block[1], win, heads)
    m.decoder_layers = nn.ModuleList([nn.ModuleList([
        nn.Sequential(_Slot(), _point_conv(embed * 2, embed)), _point_conv(embed * 2, embed),
        _lgb(embed, num_block[0], win, heads)])])
    m.tail = nn.Sequential(_Slot(), _point_conv(embed, in_channels))
    return m


class Pansharpening(nn.Module):
    """MI355X-native counterpart of reference `Pansharpening` (models/unlg_former.py:21-67)."""

    def __init__(self, cfg, logger, stage=5):
        super().__init__()
        self.in_channels = cfg.ms_chans
        self.stage = stage
        self.up_factor = 4
        C = self.in_channels
        self.D = nn.Sequential(_Slot(), _dep_conv(C, 3), _Slot(), _dep_conv(C, 3))
        self.DT = nn.Sequential(_Slot(), _dep_conv(C, 3), _Slot(), _dep_conv(C, 3))
        self.R = _point_conv(C, 1)
        self.RT = _point_conv(1, C)
        self.eta = nn.ParameterList([nn.Parameter(torch.tensor(0.1)) for _ in range(stage)])
        self.prior_module = nn.ModuleList([_lgt(C, C * 4) for _ in range(stage)])
        # execution options (not part of the reference surface)
        # 'faithful': run all K LGTs like the reference; 'live': skip the dead ones (identical results);
        # 'chained': the intended unfolding (stage i+1 consumes LGT_i's output; every parameter trains) -- NOT the reference's
        # results, opt-in only
        self.mode = 'faithful'
        self.faithful_eval = False  # True: also run the dead-stage LGTs in eval / no-grad forwards (timing the reference's work)
        self.precision = 'fp32'    # 'fp32': parity mode; 'bf16': saved / hidden FFN activations of the backward stored as bf16
        self._engine = None
        self._ddp = None           # (group,) once attach_ddp() was called: survives .to() / a rebuilt engine

    # ---- engine plumbing ----------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engine = None            # parameters were re-created (.cuda()/.to()): flat storage is stale
        return r

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_engine'] = None           # raw device handles never get pickled (torch.save of whole modules)
        st['_ddp'] = None              # ... nor process groups
        return st

    def engine(self):
        if self._engine is None or not self._engine.valid():
            self._engine = Engine(self)
            if self._ddp is not None:  # the data-parallel attachment belongs to the module, not to one Engine object
                # no broadcast here: a rebuild may happen on one rank only (a collective would hang) and must not overwrite what
                # the ranks loaded; call attach_ddp() again on EVERY rank to re-synchronise the weights
                self._engine.attach_ddp(self._ddp[0], broadcast=False, force=len(self._ddp) > 1 and self._ddp[1])
        return self._engine

    def attach_ddp(self, group=None, force=False):
        """join a torch.distributed process group (one process per GPU; backend nccl = RCCL over xGMI): rank-0 weights are
        broadcast and every train step / autograd backward all-reduces the flat gradient buffer.  A COLLECTIVE call: every rank
        of the group makes it (again after loading weights on one rank only).  Replaces the reference's nn.DataParallel wrap
        (models/base/base_model.py:91-100)."""
        self._ddp = None
        eng = self.engine()            # (re)built without an attachment
        self._ddp = (group, bool(force))
        return eng.attach_ddp(group, broadcast=True, force=force)

    def canonical_names(self):
        return canonical_names(self.in_channels, self.stage)

    def forward(self, ms, pan):
        if not ms.is_cuda:
            raise RuntimeError('lgteun_amd.Pansharpening runs on MI355X (HIP) only; there is no CPU path. '
                               'Move the module and inputs to cuda.')
        return self.engine().forward_autograd(ms, pan, training=self.training)


@MODELS.register_module()
class UnlgFormer(Base_model):
    """Runner wrapper with the reference's constructor and hooks (models/unlg_former.py:70-113)."""

    def __init__(self, cfg, logger, train_data_loader, test_data_loader0, test_data_loader1):
        super().__init__(cfg, logger, train_data_loader, test_data_loader0, test_data_loader1)
        model_cfg = cfg.get('model_cfg', dict())
        G_cfg = model_cfg.get('core_module', dict())
        self.add_module('core_module', Pansharpening(cfg=cfg, logger=logger, **G_cfg))

    def get_model_output(self, input_batch):
        input_pan = input_batch['input_pan']
        input_lr = input_batch['input_lr']
        return self.module_dict['core_module'](input_lr, input_pan)

    def train_iter(self, iter_id, input_batch, log_freq=10):
        """Same contract as the reference train_iter.  With the fused optimizer (`set_optim` default on this
        build) forward + L1 + backward + Adam run as four library calls with no per-iteration host sync
        (the reference's two `.item()` syncs, unlg_former.py:105,107, happen only every `log_freq`)."""
        G = self.module_dict['core_module']
        G_optim = self.optim_dict['core_module']
        loss_cfg = self.cfg.get('loss_cfg', {})
        w = float(loss_cfg['rec_loss'].w) if 'rec_loss' in self.loss_module else 0.0
        core = G.module if hasattr(G, 'module') else G
        if getattr(G_optim, 'is_fused_lgteun', False) and 'rec_loss' in self.loss_module and \
                self.loss_module['rec_loss'].get_type() == 'l1':
            loss_t = core.engine().train_step(input_batch['input_lr'], input_batch['input_pan'], input_batch['target'],
                                              G_optim, loss_weight=w)
            if iter_id % log_freq == 0:
                v = core.engine().global_loss()        # all ranks: under DDP `loss_t` is this rank's share of the global mean
                self.print_train_log(iter_id, dict(rec_loss=v / w if w else 0.0, full_loss=v), log_freq)
            return
        output = G(input_batch['input_lr'], input_batch['input_pan'])
        loss_g = 0
        loss_res = dict()
        if 'rec_loss' in self.loss_module:
            rec_loss = self.loss_module['rec_loss'](out=output, gt=input_batch['target'])
            loss_g = loss_g + rec_loss * w
            loss_res['rec_loss'] = rec_loss.item()
        loss_res['full_loss'] = loss_g.item()
        G_optim.zero_grad()
        loss_g.backward()
        G_optim.step()
        self.print_train_log(iter_id, loss_res, log_freq)
