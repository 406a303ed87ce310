"""-m gpu: PAN sizes that are not square powers of two (full-resolution 400x400 scenes, rectangular crops; any
multiples of 16 up to 1024).  Only the FFT mixer changes path (Bluestein lines through the radix-2 passes, k_fft.hip); every
other kernel already takes h and w -- this file is what proves that, against the oracle (torch.fft handles any size)."""
import numpy as np
import pytest
import torch

from helpers import det_params, rel_l2
from oracle import detweights as dw
from oracle import lgteun_oracle as orc

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(autouse=True)
def canonical_real_bins(monkeypatch):
    """non-power-of-two sizes: the sign of the zero imaginary part torch's FFT leaves in the four purely-real bins -- hence
    angle() = +pi or -pi where they are negative -- depends on the host CPU (oracle/lgteun_oracle.py); pin the +0 convention"""
    monkeypatch.setattr(orc, 'CANONICAL_REAL_BINS', True)


@pytest.mark.parametrize('C,H,W', [(4, 48, 48), (4, 32, 64), (8, 64, 32), (4, 80, 48), (4, 16, 48), (4, 208, 176)])
def test_global_mixer_any_size_vs_oracle(C, H, W):
    """the FFT mixer alone at both levels: forward, input gradient and the four parameter gradients"""
    from gpu_helpers import Ops, make_module
    net = make_module(C, 1)
    ops = Ops(net, H, W)
    P = det_params(C, 1, requires_grad=True)
    E = 4 * C
    rng = np.random.default_rng(H * 1000 + W)
    pre = 'prior_module.0.'
    for blk, bp, (h, w, e) in ((0, pre + 'encoder_layers.0.0.blocks.0.', (H, W, E)), (2, pre + 'bottleneck.blocks.0.', (H // 2, W // 2, 2 * E))):
        feat = T(rng.standard_normal((2, h, w, e)).astype(np.float32))
        feat[1, ..., e // 2:] -= 1.5                       # negative-DC planes: pins the pi branch of angle()
        y = orc.layer_norm(feat, P[bp + '0.fn.norm.weight'], P[bp + '0.fn.norm.bias']).detach()
        xin = y[..., e // 2:].clone().requires_grad_(True)
        want = orc.global_mixer(P, bp + '0.fn.fn.global_mixer.', xin).permute(0, 3, 1, 2)
        got = ops.block(0, blk, 0, feat.cuda()).cpu()
        assert rel_l2(got, want.detach()) < 2e-4, (blk, rel_l2(got, want.detach()))
        dy = T(rng.standard_normal(tuple(want.shape)).astype(np.float32))
        for v in P.values():
            v.grad = None
        want.backward(dy)
        dx, grads = ops.block_bwd(0, blk, 0, feat.cuda(), dy.cuda())
        assert rel_l2(dx.cpu(), xin.grad.permute(0, 3, 1, 2)) < 2e-3
        for n in ('conv_amp.0.weight', 'conv_amp.0.bias', 'conv_pha.0.weight', 'conv_pha.0.bias'):
            name = bp + '0.fn.fn.global_mixer.' + n
            assert rel_l2(ops.grad_of(grads, name).cpu(), P[name].grad) < 2e-3, name


CASES = [(4, 2, 48, 48, 2), (4, 1, 32, 64, 3), (8, 1, 64, 32, 2), (4, 2, 80, 48, 1), (4, 1, 16, 48, 2)]   # (C, K, H, W, B)


@pytest.mark.parametrize('C,K,H,W,B', CASES)
def test_whole_net_any_size_vs_oracle(C, K, H, W, B):
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    ms, pan, gt = (T(a) for a in dw.make_inputs(B, C, H // 4, W // 4, seed=500 + H + W, kind='smooth'))
    net = make_module(C, K)
    with torch.no_grad():
        y = net(ms.cuda(), pan.cuda()).cpu()
    P = det_params(C, K, requires_grad=True)
    want = orc.forward(P, ms, pan, K, mode='faithful')
    assert y.shape == (B, C, H, W)
    assert rel_l2(y, want.detach()) < 1e-3
    loss_ref = orc.l1_loss(want, gt)
    loss_ref.backward()
    opt = FusedAdam(net.parameters(), lr=0.0)
    opt.dropout = False
    eng = net.engine()
    loss = float(eng.train_step(ms.cuda(), pan.cuda(), gt.cuda(), opt).item())
    assert abs(loss - float(loss_ref.detach())) < 1e-4 * max(1.0, abs(float(loss_ref.detach())))
    num = den = 0.0
    for i in eng.live_idx:
        n, o, p = eng.names[i], eng.offsets[i], eng.params[i]
        got = eng.gflat[o:o + p.numel()].view(p.shape).cpu().double()
        ref = P[n].grad.double()
        num += float(((got - ref) ** 2).sum())
        den += float((ref ** 2).sum())
    assert (num / den) ** 0.5 < 5e-3, (num / den) ** 0.5


def test_full_resolution_scene_400():
    """the reference's test_full_res scenes (README.md:88; 100x100 MS / 400x400 PAN): forward vs the oracle, and one train step runs"""
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    C, K, H, B = 4, 2, 400, 1
    ms, pan, gt = (T(a) for a in dw.make_inputs(B, C, H // 4, H // 4, seed=4, kind='smooth'))
    net = make_module(C, K)
    with torch.no_grad():
        y = net(ms.cuda(), pan.cuda()).cpu()
        want = orc.forward(det_params(C, K), ms, pan, K, mode='live')
    assert rel_l2(y, want) < 1e-3
    opt = FusedAdam(net.parameters(), lr=1e-3)
    loss = float(net.engine().train_step(ms.cuda(), pan.cuda(), gt.cuda(), opt).item())
    assert np.isfinite(loss)


def test_rejected_sizes():
    from gpu_helpers import make_module
    net = make_module(4, 1)
    for h, w in ((6, 8), (8, 260)):           # PAN 24x32: not a multiple of 16; PAN 32x1040: beyond the 1024 limit
        ms, pan, _ = (T(a).cuda() for a in dw.make_inputs(1, 4, h, w, seed=1, kind='smooth'))
        with pytest.raises(RuntimeError, match='plan_create'):
            net(ms, pan)
