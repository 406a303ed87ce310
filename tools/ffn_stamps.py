"""GPU probe (diagnostic build, tools/build_stamps.sh): phase stamps of the fused FFN kernel's third step, workgroup 0, per wave.
   LGTEUN_HIP_LIB=$PWD/lgteun_amd/_lgteun_hip_stamps.so python tools/ffn_stamps.py"""
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
x = torch.from_numpy(np.random.default_rng(0).standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
for _ in range(5):
    ops.block(0, 0, 2, x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
L = ops.lib
L.lg_debug_ffn_stamps.restype = ctypes.c_int
assert L.lg_debug_ffn_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(4, 32).astype(np.int64)
names = {0: 'step start', 1: 'barrier(alias)', 2: 'LN0 stored', 3: 'barrier'}
for c in range(3):
    names.update({4 + 4 * c: f'c{c} GEMM1+GELU+split', 5 + 4 * c: f'c{c} barrier', 6 + 4 * c: f'c{c} GEMM2+LN+ring', 7 + 4 * c: f'c{c} barrier'})
for ch in range(2):
    names.update({16 + 3 * ch: f'P2 row{ch} dw+GELU+split', 17 + 3 * ch: f'P2 row{ch} GEMM3', 18 + 3 * ch: f'P2 row{ch} epilogue'})
order = [0, 1, 2, 3] + [4 + 4 * c + k for c in range(3) for k in range(4)] + [16, 17, 18, 19, 20, 21]
print('phase'.ljust(28), *[f'wave{w}'.rjust(8) for w in range(4)], '   (cycles spent in the phase; s_memtime ticks)')
prev = st[:, 0].copy()
for i in order[1:]:
    d = st[:, i] - prev
    print(names[i].ljust(28), *[str(int(v)).rjust(8) for v in d])
    prev = st[:, i].copy()
print('step total'.ljust(28), *[str(int(v)).rjust(8) for v in st[:, 21] - st[:, 0]])
