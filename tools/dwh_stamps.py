"""GPU probe (diagnostic variant: bash tools/mkvariant.sh dwh_stamps k_ffn_dwbwd_h.hip -DLG_STAMPS): phase stamps of k_ffn_dw_bwd_h, means over the workgroups.
   LGTEUN_HIP_LIB=$PWD/build_variants/dwh_stamps.so python tools/dwh_stamps.py"""
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
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
dy = torch.from_numpy(rng.standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
for _ in range(3):
    ops.block_bwd(0, 0, 2, x, dy)
torch.cuda.synchronize()
n = 1024 * 4 * 10 * 8
buf = (ctypes.c_ulonglong * n)()
L = ops.lib
L.lg_debug_dwh_stamps.restype = ctypes.c_int
assert L.lg_debug_dwh_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 10, 8).astype(np.int64)
names = ['barrier A + h2 ring stores + barrier B', 'prefetch issue + halo pass', 'barrier C (dh3 complete)', 'spatial phase']
print('s_memtime ticks, mean over 1024 workgroups (2 halves x 512 strips); columns = waves')
for si in (1, 3, 6):
    print(f' step {si}')
    for k, nm in enumerate(names):
        d = (st[:, :, si, k + 1] - st[:, :, si, k]).mean(axis=0)
        print('  ' + nm.ljust(42), *[f'{v:9.1f}' for v in d])
    print('  ' + 'step total (start of this -> start of next)'.ljust(42), *[f'{v:9.1f}' for v in (st[:, :, si + 1, 0] - st[:, :, si, 0]).mean(axis=0)])
print(' eight steps', *[f'{v:9.1f}' for v in (st[:, :, 7, 4] - st[:, :, 0, 0]).mean(axis=0)])
