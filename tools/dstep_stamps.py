"""GPU probe (diagnostic build, tools/build_stamps.sh): phase stamps of the one-launch data step (k_dstep.hip), every workgroup and wave.
   LGTEUN_HIP_LIB=$PWD/lgteun_amd/_lgteun_hip_stamps.so python tools/dstep_stamps.py [fwd|bwd] [C]"""
import ctypes
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module

which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B, N = 32, 128
ops = Ops(make_module(C, 2), N, N)
rng = np.random.default_rng(0)
T = lambda *s: torch.from_numpy(rng.uniform(0, 1, s).astype(np.float32)).cuda()
z, ms, pan, dy = T(B, C, N, N), T(B, C, N // 4, N // 4), T(B, 1, N, N), T(B, C, N, N)
run = (lambda: ops.data_step(1, z, ms, pan)) if which == 'fwd' else (lambda: ops.data_step_bwd(1, z, ms, pan, dy))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print(f'one {which} data step by torch events (includes the op wrapper): {1e3 * e0.elapsed_time(e1):.1f} us')
NW, NS = 16, 32
buf = (ctypes.c_ulonglong * (512 * NW * NS))()
f = ops.lib.lg_debug_ds_stamps
f.restype = ctypes.c_int
assert f(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, NW, NS).astype(np.int64)[:B * C]
last = int(np.max(np.nonzero(st[0, 0])[0]))  # LDS stamp slots are zeroed at kernel start
print(f'{B * C} workgroups, {last} phases; s_memtime ticks (shader cycles, ~2.3 GHz) per phase: mean over workgroups of (slowest wave end - slowest wave start)')
t0 = st[:, :, 0].min()
for i in range(1, last + 1):
    d = st[:, :, i].max(axis=1) - st[:, :, i - 1].max(axis=1)
    print(f'  phase {i:2d}: {d.mean():8.1f}   (min {d.min()}, max {d.max()})')
tot = st[:, :, last].max(axis=1) - st[:, :, 0].min(axis=1)
print(f'  workgroup lifetime {tot.mean():.1f} ticks (min {tot.min()}, max {tot.max()}); first start -> last end {st[:, :, last].max() - t0}; start spread {st[:, :, 0].min(axis=1).max() - t0}')
