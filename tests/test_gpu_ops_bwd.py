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


@pytest.mark.parametrize('H,B', [(128, 2), (128, 33), (64, 3), (32, 2), (16, 2)])    # (8 x 8 planes: level 2 of the 32 x 32 whole-net goldens)
def test_real_input_fft_mixer_agrees_with_the_complex_row_kernels(H, B, monkeypatch):
    """k_fftmix_r / k_fftmix_bwd_r (round 5: rows as n/2-point transforms of packed reals, compact n x (n/2 + 1) plane, wave-uniform twiddles
    from constant memory; B = 33 at 128 x 128 = 264 planes takes the two-workgroups-per-CU instance) against the complex-row kernels
    (LG_FFT=full) on the same random features: the global mixer's forward output, and its backward (dx and the four parameter gradients)
    from the same upstream gradient.  Two fp32 evaluations of the same transforms: agreement to a few 1e-7 of the output's norm; the
    parameter gradients are sums of ~1e5 terms of both signs (tolerance relative to the sum of magnitudes would be 1e-7: 1e-4 of the value)."""
    from gpu_helpers import Ops, make_module
    rng = np.random.default_rng(100 + H + B)
    feat = T(rng.standard_normal((B, H, H, 16)).astype(np.float32)).cuda()
    dy = T(rng.standard_normal((B, 8, H, H)).astype(np.float32)).cuda()
    res = {}
    for kind in ('real', 'full'):
        if kind == 'full':
            monkeypatch.setenv('LG_FFT', 'full')          # read once per plan: a fresh module builds a fresh plan
        else:
            monkeypatch.delenv('LG_FFT', raising=False)
        ops = Ops(make_module(4, 1), H, H)
        y = ops.block(0, 0, 0, feat)
        dx, grads = ops.block_bwd(0, 0, 0, feat, dy)
        pre = 'prior_module.0.encoder_layers.0.0.blocks.0.0.fn.fn.global_mixer.'
        pg = {k: ops.grad_of(grads, pre + k).double().cpu() for k in ('conv_amp.0.weight', 'conv_amp.0.bias', 'conv_pha.0.weight', 'conv_pha.0.bias')}
        res[kind] = (y.double().cpu(), dx.double().cpu(), pg)
    monkeypatch.delenv('LG_FFT', raising=False)
    (y0, dx0, pg0), (y1, dx1, pg1) = res['real'], res['full']
    assert float(y1.norm()) > 0 and float(dx1.norm()) > 0
    assert float((y0 - y1).norm()) <= 2e-6 * float(y1.norm()), float((y0 - y1).norm() / y1.norm())
    assert float((dx0 - dx1).norm()) <= 1e-5 * float(dx1.norm()), float((dx0 - dx1).norm() / dx1.norm())
    for k in pg0:
        assert float((pg0[k] - pg1[k]).norm()) <= 1e-4 * float(pg1[k].norm()) + 1e-9, (k, float((pg0[k] - pg1[k]).norm()), float(pg1[k].norm()))
