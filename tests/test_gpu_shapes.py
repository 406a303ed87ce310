"""-m gpu: shape sweep of the whole path against the oracle (same name-hashed weights, same seeded inputs): smallest plane the
plan accepts (16x16 PAN: one 8x8 window at level 1), odd and prime batch sizes (ragged last tiles / window groups / persistent
grids), K = 1 and 3, both band counts.  Forward: north_star's 1e-3 relative gate.  Gradients: global relative L2 over all
live tensors against the oracle's autograd (fp32 CPU), and the dead stages' None / untouched-slot behaviour."""
import numpy as np
import pytest
import torch

from helpers import det_params, rel_l2
from oracle import detweights as dw
from oracle import lgteun_oracle as orc

pytestmark = pytest.mark.gpu

T = torch.from_numpy

CASES = [  # (C, K, PAN, B)
    (4, 1, 16, 1), (4, 3, 16, 5), (4, 2, 32, 3), (4, 1, 64, 1), (4, 2, 64, 7), (8, 1, 16, 3), (8, 2, 32, 5), (8, 1, 64, 2),
]


@pytest.mark.parametrize('C,K,H,B', CASES)
def test_forward_and_gradients_vs_oracle(C, K, H, B):
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    ms, pan, gt = (T(a) for a in dw.make_inputs(B, C, H // 4, H // 4, seed=100 + H + B, kind='smooth'))
    net = make_module(C, K)
    net.faithful_eval = True            # the first forward really runs the K-1 dead-stage LGTs
    with torch.no_grad():
        y = net(ms.cuda(), pan.cuda()).cpu()
        net.mode = 'live'
        y_live = net(ms.cuda(), pan.cuda()).cpu()
        net.mode = 'faithful'
    P = det_params(C, K, requires_grad=True)
    want = orc.forward(P, ms, pan, K, mode='faithful')
    assert torch.equal(y, y_live)                                   # dead-stage LGTs never reach the output (D3)
    assert rel_l2(y, want.detach()) < 1e-3
    loss_ref = orc.l1_loss(want, gt)
    loss_ref.backward()
    opt = FusedAdam(net.parameters(), lr=0.0)
    opt.dropout = False
    eng = net.engine()
    loss = float(eng.train_step(ms.cuda(), pan.cuda(), gt.cuda(), opt).item())
    assert abs(loss - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    num = den = 0.0
    live = set()
    for i in eng.live_idx:
        n, o, p = eng.names[i], eng.offsets[i], eng.params[i]
        live.add(n)
        got = eng.gflat[o:o + p.numel()].view(p.shape).cpu().double()
        ref = P[n].grad.double()
        num += float(((got - ref) ** 2).sum())
        den += float((ref ** 2).sum())
    assert (num / den) ** 0.5 < 5e-3, (num / den) ** 0.5
    assert {n for n, v in P.items() if v.grad is not None} == live  # the oracle's autograd touches exactly the live set
    if K > 1:
        a, b = eng.live_ranges[0][1], eng.live_ranges[1][0]
        assert float(eng.gflat[a:b].abs().max()) == 0.0
