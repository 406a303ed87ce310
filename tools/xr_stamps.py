"""GPU probe (diagnostic variant: bash tools/mkvariant.sh xr_stamps k_ffn_xr.hip -DLG_STAMPS): phase stamps of k_ffn_xr, every step of every workgroup's
first strip, means over the workgroups.   LGTEUN_HIP_LIB=$PWD/build_variants/xr_stamps.so python tools/xr_stamps.py"""
import ctypes
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module

net = make_module(4, 1)
ops = Ops(net, 128, 128)
NB_ = int(os.environ.get("XR_B", "32"))
x = torch.from_numpy(np.random.default_rng(0).standard_normal((NB_, 128, 128, 16)).astype(np.float32)).cuda()
for _ in range(5):
    ops.block(0, 0, 2, x)
torch.cuda.synchronize()
n = 512 * 4 * 10 * 8
buf = (ctypes.c_ulonglong * n)()
L = ops.lib
L.lg_debug_xr_stamps.restype = ctypes.c_int
assert L.lg_debug_xr_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 10, 8).astype(np.int64)
names = ['halo pair (2 blocks)', 'ninth block (one wave in turn)', 'barrier 1 (ring complete)', 'spatial phase (2 rows)', 'barrier 2 (ring free)']
print('s_memtime ticks (100 MHz constant clock x ~21 = core cycles at 2.1 GHz), mean over 512 workgroups; columns = waves')
for si in range(8):
    print(f' step {si}')
    for k, nm in enumerate(names):
        d = (st[:, :, si, k + 1] - st[:, :, si, k]).mean(axis=0)
        print('  ' + nm.ljust(34), *[f'{v:9.1f}' for v in d])
    print('  ' + 'step total'.ljust(34), *[f'{v:9.1f}' for v in (st[:, :, si, 5] - st[:, :, si, 0]).mean(axis=0)])
print(' kernel start -> tables staged -> prologue done:', *[f'{v:9.1f}' for v in (st[:, :, 9, 1] - st[:, :, 9, 0]).mean(axis=0)], '|', *[f'{v:9.1f}' for v in (st[:, :, 9, 2] - st[:, :, 9, 1]).mean(axis=0)])
print(' workgroup start spread:', int(st[:, :, 9, 0].max() - st[:, :, 9, 0].min()), ' first start -> last step end:', int(st[:, :, 7, 5].max() - st[:, :, 9, 0].min()))
tot = (st[:, :, 7, 5] - st[:, :, 0, 0]).mean(axis=0)
print(' eight steps'.ljust(37), *[f'{v:9.1f}' for v in tot])
print(' kernel span over workgroups (first stamp -> last stamp):', int(st[:, :, 7, 5].max() - st[:, :, 0, 0].min()))
# per-workgroup timeline: when do workgroups start and end relative to the first one?
t0 = st[:, 0, 9, 0]
ok = t0 > 0
base = t0[ok].min()
start = (t0 - base)[ok]
end = (st[:, 0, 7, 5] - base)[ok]
print(f' workgroups with stamps: {int(ok.sum())}   start (ticks after the first): p10 {np.percentile(start, 10):.0f}  p50 {np.percentile(start, 50):.0f}  p90 {np.percentile(start, 90):.0f}  max {start.max():.0f}')
print(f'   end of step 7: p10 {np.percentile(end, 10):.0f}  p50 {np.percentile(end, 50):.0f}  p90 {np.percentile(end, 90):.0f}  max {end.max():.0f}')
for si in range(8):
    a = (st[:, 0, si, 0] - base)[ok]
    print(f'   step {si} begins: p10 {np.percentile(a, 10):.0f}  p50 {np.percentile(a, 50):.0f}  p90 {np.percentile(a, 90):.0f}')
# are the two workgroups of a CU treated alike?  lifetime (first step's begin -> last step's end) per workgroup, and the per-step durations of the fast and the slow half
life = (st[:, 0, 7, 5] - st[:, 0, 0, 0])[ok]
print(f' workgroup lifetime over 8 steps: p5 {np.percentile(life, 5):.0f}  p25 {np.percentile(life, 25):.0f}  p50 {np.percentile(life, 50):.0f}  p75 {np.percentile(life, 75):.0f}  p95 {np.percentile(life, 95):.0f}')
med = np.median(life)
fast, slow = life <= med, life > med
stepd = (st[:, 0, :8, 5] - st[:, 0, :8, 0])[ok]
print('   step durations, faster half of the workgroups:', *[f'{v:8.0f}' for v in stepd[fast].mean(axis=0)])
print('   step durations, slower half of the workgroups:', *[f'{v:8.0f}' for v in stepd[slow].mean(axis=0)])
print('   histogram of lifetimes (ticks):', np.histogram(life, bins=10))
# does dispatch order decide who is the older workgroup of a CU?  lifetime against blockIdx
idx = np.arange(512)[ok]
lo, hi = life[idx < 256], life[idx >= 256]
print(f' lifetime of workgroups 0..255: mean {lo.mean():.0f} (min {lo.min():.0f}, max {lo.max():.0f});  256..511: mean {hi.mean():.0f} (min {hi.min():.0f}, max {hi.max():.0f})')
print(f' even blockIdx: mean {life[idx % 2 == 0].mean():.0f};  odd: {life[idx % 2 == 1].mean():.0f}')
