"""ReconstructionLoss + get_loss_module -- mirror of reference models/base/losses.py:19-40,222-249 restricted to
what configs/unlg_former.py:88-90 uses (rec_loss, type l1/l2).  Adversarial/QNR/MI losses are out of scope."""
import torch.nn as nn


class ReconstructionLoss(nn.Module):
    def __init__(self, cfg, logger, loss_type='l1'):
        super().__init__()
        self.cfg = cfg
        self.loss_type = loss_type
        if loss_type == 'l1':
            self.loss = nn.L1Loss()
        elif loss_type == 'l2':
            self.loss = nn.MSELoss()
        else:
            if logger is not None:
                logger.error(f'No such type of ReconstructionLoss: "{loss_type}"')
            raise SystemExit(f'No such type of ReconstructionLoss: "{loss_type}"')

    def get_type(self):
        return self.loss_type

    def forward(self, out, gt):
        return self.loss(out, gt)


def get_loss_module(full_cfg, logger):
    loss_cfg = full_cfg.get('loss_cfg')
    loss_module = dict()
    for loss_name in loss_cfg:
        cfg = loss_cfg[loss_name]
        if 'rec_loss' in loss_name:
            if abs(cfg.w - 0) > 1e-8:
                loss_module[loss_name] = ReconstructionLoss(cfg, logger, loss_type=cfg.type)
        else:
            raise SystemExit(f'loss "{loss_name}" is outside the LGTEUN hot path of this build (only rec_loss)')
    return loss_module
