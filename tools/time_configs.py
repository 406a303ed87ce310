"""Train-step timing of the BASELINE.json / SURVEY 8d configurations on one GPU (fp32, faithful and live, dropout on).
These are parity-test configurations, not bench lines (bench.py measures configs[1]); recorded in DESIGN.md section 4.
usage (GPU box): python tools/time_configs.py [--steps 10]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--only', default='', help='substring of the configuration name (e.g. c3)')
    args = ap.parse_args()
    import lgteun_amd
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    cfgs = [('c2', 4, 128, 4, 32), ('c3', 8, 128, 4, 32), ('c5 (C=8)', 8, 256, 8, 16), ('c5 with C=4', 4, 256, 8, 16), ('512^2 PAN', 4, 512, 4, 4), ('400^2 PAN', 4, 400, 4, 4)]
    for name, C, H, K, B in cfgs:
        if args.only and args.only not in name:
            continue
        for mode in ('faithful', 'live', 'chained'):
            torch.cuda.reset_peak_memory_stats()
            net = make_module(C, K)
            net.mode = mode
            net.train()
            eng = net.engine()
            opt = FusedAdam(net.parameters(), lr=1.5e-3)
            g = torch.Generator().manual_seed(19971118)

            def dn(*shape):
                return (torch.randint(0, 2048, shape, generator=g).float() / 2047.5).cuda()
            ms, pan, gt = dn(B, C, H // 4, H // 4), dn(B, 1, H, H), dn(B, C, H, H)
            for _ in range(3):
                eng.train_step(ms, pan, gt, opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.train_step(ms, pan, gt, opt)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            print(f'{name:12s} C={C} PAN {H}x{H} K={K} B={B:3d} {mode:8s}: {dt * 1e3:8.2f} ms/step  {B / dt:8.1f} pairs/s  '
                  f'peak mem {torch.cuda.max_memory_allocated() / 2**30:6.2f} GiB', flush=True)
            del net, eng, opt
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()


if __name__ == '__main__':
    main()
