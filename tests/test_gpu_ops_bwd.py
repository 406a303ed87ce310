"""-m gpu: backward of the path's two top-level pieces in isolation through their own C-ABI entries (one fwd + one bwd
entry per fused unit), against torch autograd over the fp64 oracle: the data step (clamp-aware adjoints of the polyphase
resamplers, depthwise transposes, R / RT, eta) and one whole LGT (embed / down / up+fusion / tail backward besides the blocks)."""
import numpy as np
import pytest
import torch

from helpers import det_params, rel_l2
from oracle import lgteun_oracle as orc

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize('C,H,W', [(4, 32, 32), (8, 48, 16), (4, 128, 128), (8, 128, 128), (4, 64, 64), (8, 64, 64)])
def test_data_step_backward_vs_oracle(C, H, W):
    from gpu_helpers import Ops, make_module
    K, B, stage = 2, 2, 1
    net = make_module(C, K)
    ops = Ops(net, H, W)
    rng = np.random.default_rng(C * 1000 + H)
    z = rng.uniform(0, 1, (B, C, H, W)).astype(np.float32)
    ms = rng.uniform(0, 1, (B, C, H // 4, W // 4)).astype(np.float32)
    pan = rng.uniform(0, 1, (B, 1, H, W)).astype(np.float32)
    dy = rng.standard_normal((B, C, H, W)).astype(np.float32)
    P = det_params(C, K, dtype=torch.float64, requires_grad=True)
    zz = T(z).double().requires_grad_(True)
    out = orc.data_step(P, zz, T(ms).double(), T(pan).double(), P[f'eta.{stage}'])
    (out * T(dy).double()).sum().backward()
    dz, grads = ops.data_step_bwd(stage, T(z).cuda(), T(ms).cuda(), T(pan).cuda(), T(dy).cuda())
    assert rel_l2(dz.cpu(), zz.grad) < 2e-6
    live = [n for n, v in P.items() if v.grad is not None]
    assert sorted(live) == sorted([n for n in P if n.split('.')[0] in ('D', 'DT', 'R', 'RT')] + [f'eta.{stage}'])
    for n in live:
        assert rel_l2(ops.grad_of(grads, n).cpu(), P[n].grad) < 2e-5, n
    touched = torch.zeros_like(grads, dtype=torch.bool)
    for n in live:
        i = ops.eng.names.index(n)
        touched[ops.eng.offsets[i]:ops.eng.offsets[i] + ops.eng.params[i].numel()] = True
    assert float(grads[~touched].abs().max()) == 0.0           # nothing else is written


@pytest.mark.parametrize('C,H', [(4, 32), (8, 32), (4, 64)])
def test_lgt_backward_vs_oracle(C, H):
    from gpu_helpers import Ops, make_module
    K, B, stage = 2, 2, 1
    net = make_module(C, K)
    ops = Ops(net, H, H)
    rng = np.random.default_rng(C * 100 + H)
    z = rng.uniform(0, 1, (B, C, H, H)).astype(np.float32)
    dy = rng.standard_normal((B, C, H, H)).astype(np.float32)
    P = det_params(C, K, dtype=torch.float64, requires_grad=True)
    zz = T(z).double().requires_grad_(True)
    pre = f'prior_module.{stage}.'
    out = orc.lgt(P, pre, zz)
    (out * T(dy).double()).sum().backward()
    dz, grads = ops.lgt_bwd(stage, T(z).cuda(), T(dy).cuda())
    assert rel_l2(dz.cpu(), zz.grad) < 2e-4, rel_l2(dz.cpu(), zz.grad)
    names = [n for n in P if n.startswith(pre)]
    assert len(names) == 119 and all(P[n].grad is not None for n in names)
    num = sum(float(((ops.grad_of(grads, n).cpu().double() - P[n].grad) ** 2).sum()) for n in names)
    den = sum(float((P[n].grad ** 2).sum()) for n in names)
    assert (num / den) ** 0.5 < 5e-4, (num / den) ** 0.5
    # the pieces no other per-op entry reaches, one by one
    for n in ('patch_embed.proj.0.weight', 'patch_embed.proj.1.weight', 'patch_embed.norm.weight', 'encoder_layers.0.1.1.weight',
              'decoder_layers.0.0.1.weight', 'decoder_layers.0.1.weight', 'decoder_layers.0.1.bias', 'tail.1.weight', 'tail.1.bias'):
        assert rel_l2(ops.grad_of(grads, pre + n).cpu(), P[pre + n].grad) < 2e-3, n
