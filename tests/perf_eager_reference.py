"""NOT a test (no test_ prefix; pytest does not collect it): what the reference's graph costs on the SAME MI355X in eager
PyTorch-ROCm, i.e. what a user of the reference gets on this GPU today.  The reference itself cannot travel to the GPU box, so
the graph is the ORACLE (pinned against the reference's own outputs in test_oracle_golden.py) with its hand-written primitives
swapped for the ATen ops the reference calls -- F.interpolate(bicubic) (basic_module_unformer_v2.py:21-34), F.conv2d dense /
depthwise (:13-18), F.layer_norm (LGT.py:58), F.gelu (LGT.py:97-99); matmul / softmax / torch.fft are ATen already -- checked
here against the unpatched oracle before timing.  One train step = forward of all K stages (faithful, like
unlg_former.py:56-67) + L1 + backward + torch.optim.Adam.  fp32, no dropout (slightly in the baseline's favour), BASELINE
configs[1] by default.  The number goes into DESIGN.md section 4 as a baseline beside `value`; nothing in the product or in
bench.py's `value` uses this path.

    python tests/perf_eager_reference.py [--batch 32] [--steps 10] [--C 4] [--pan 128] [--K 4]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from helpers import det_params  # noqa: E402
from oracle import lgteun_oracle as orc  # noqa: E402


def use_aten_ops():
    import torch.nn.functional as F
    orc.resample = lambda x, s: x if s == 1 else F.interpolate(x, scale_factor=s, mode='bicubic', align_corners=False,
                                                               recompute_scale_factor=False)
    orc.point_conv = lambda x, w, b: F.conv2d(x, w, b)
    orc.dep_conv = lambda x, w, b: F.conv2d(x, w, b, padding=w.shape[-1] // 2, groups=x.shape[1])
    orc.layer_norm = lambda x, g, b, eps=1e-5: F.layer_norm(x, (x.shape[-1],), g, b, eps)
    orc.gelu = F.gelu


def check_equivalence():
    """the ATen-op graph equals the hand-written oracle (CPU, small case)"""
    from oracle import detweights as dw
    P = det_params(4, 2)
    ms, pan, _ = (torch.from_numpy(a) for a in dw.make_inputs(2, 4, 8, 8, seed=3, kind='smooth'))
    want = orc.forward(P, ms, pan, 2, mode='faithful')
    use_aten_ops()
    got = orc.forward(P, ms, pan, 2, mode='faithful')
    err = float((got - want).norm() / want.norm())
    assert err < 5e-5, err
    return err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--C', type=int, default=4)
    ap.add_argument('--pan', type=int, default=128)
    ap.add_argument('--K', type=int, default=4)
    ap.add_argument('--mode', default='faithful', choices=['faithful', 'live'])
    ap.add_argument('--forward-only', action='store_true', help='eval forward under no_grad instead of a train step')
    a = ap.parse_args()
    print(f'ATen-op graph vs hand-written oracle (CPU, C=4 K=2 PAN 32): rel L2 {check_equivalence():.2e}', flush=True)
    dev = torch.device('cuda:0')
    P = {k: v.to(dev).requires_grad_(True) for k, v in det_params(a.C, a.K).items()}
    g = torch.Generator().manual_seed(19971118)

    def dn(*shape):
        return (torch.randint(0, 2048, shape, generator=g).float() / 2047.5).to(dev)
    B, H = a.batch, a.pan
    ms, pan, gt = dn(B, a.C, H // 4, H // 4), dn(B, 1, H, H), dn(B, a.C, H, H)
    opt = torch.optim.Adam(list(P.values()), lr=1.5e-3)

    def step():
        if a.forward_only:
            with torch.no_grad():
                return orc.forward(P, ms, pan, a.K, mode=a.mode)
        opt.zero_grad(set_to_none=True)
        loss = orc.l1_loss(orc.forward(P, ms, pan, a.K, mode=a.mode), gt)
        loss.backward()
        opt.step()
        return loss
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f'eager PyTorch-ROCm, reference op graph ({a.mode}, fp32, no dropout, {"eval forward" if a.forward_only else "train step"}) C={a.C} PAN {H}x{H} K={a.K} B={B}: {dt * 1e3:.2f} ms/step '
          f'{B / dt:.1f} pairs/s  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  torch {torch.__version__}', flush=True)


if __name__ == '__main__':
    main()
