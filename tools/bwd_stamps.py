"""GPU probe (diagnostic build, tools/build_stamps.sh): phase stamps of the two e = 16 FFN backward kernels, workgroup 0, per wave.
   LGTEUN_HIP_LIB=$PWD/lgteun_amd/_lgteun_hip_stamps.so python tools/bwd_stamps.py"""
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
L = ops.lib


def show(fn, names, last):
    buf = (ctypes.c_ulonglong * 64)()
    f = getattr(L, fn)
    f.restype = ctypes.c_int
    assert f(buf) == 0
    st = np.array(buf, dtype=np.uint64).reshape(4, 16).astype(np.int64)
    print(fn, '  (s_memtime ticks spent in the phase, per wave of workgroup 0)')
    prev = st[:, 0].copy()
    for i in range(1, last + 1):
        d = st[:, i] - prev
        print('  ' + names[i].ljust(34), *[str(int(v)).rjust(8) for v in d])
        prev = st[:, i].copy()
    print('  ' + 'total'.ljust(34), *[str(int(v)).rjust(8) for v in st[:, last] - st[:, 0]])


show('lg_debug_kb_stamps', {1: 'loader: split dh2, LN(x), issue next', 2: 'barrier', 3: 'GEMM phase (4 pixel blocks)', 4: 'barrier', 5: 'LayerNorm backward + dx store'}, 5)
show('lg_debug_ka_stamps', {1: 'h3 fetch c0 + dy store c0', 2: 'barrier', 3: 'chunk 0', 4: 'barrier', 5: 'chunk 1', 6: 'barrier', 7: 'chunk 2 (+ h2 requests)', 8: 'barrier',
                            9: 'taps reload', 10: 'P2: 8 items (dw^T, tap gradients, dh2 store)'}, 10)
