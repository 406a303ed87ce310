"""The loss surface the LGTEUN runner needs (reference models/base/losses.py: `ReconstructionLoss` :19-40 and the
`get_loss_module` factory :222-249, as used by configs/unlg_former.py:88-90 -- one `rec_loss` entry of type l1 / l2 with weight w).
The adversarial / QNR / mutual-information losses of the comparison methods are outside this build.
The fused train step does not call this module: it computes L1 + its gradient in `lg_l1_loss`; this is the autograd-path /
torch-optimizer route and what `get_type()` callers see."""
import torch.nn as nn

_CRITERIA = {'l1': nn.L1Loss, 'l2': nn.MSELoss}


class ReconstructionLoss(nn.Module):
    """mean |out - gt| (l1) or mean (out - gt)^2 (l2); unknown types end the run like the reference does (SystemExit)"""

    def __init__(self, cfg, logger, loss_type='l1'):
        super().__init__()
        if loss_type not in _CRITERIA:
            msg = f'No such type of ReconstructionLoss: "{loss_type}"'
            if logger is not None:
                logger.error(msg)
            raise SystemExit(msg)
        self.cfg, self.loss_type = cfg, loss_type
        self.loss = _CRITERIA[loss_type]()

    def get_type(self):
        return self.loss_type

    def forward(self, out, gt):
        return self.loss(out, gt)


def get_loss_module(full_cfg, logger):
    """{name: module} for every configured loss whose weight is non-zero; only reconstruction losses exist on this path"""
    modules = {}
    table = full_cfg.get('loss_cfg') or {}
    for name in table:
        if 'rec_loss' not in name:
            raise SystemExit(f'loss "{name}" is outside the LGTEUN hot path of this build (only rec_loss)')
        cfg = table[name]                 # attribute-style entry (mmcv Config / compat.Config) or a plain dict
        field = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        if float(field('w')) != 0.0:
            modules[name] = ReconstructionLoss(cfg, logger, loss_type=field('type'))
    return modules
