"""GPU probe (diagnostic variant: bash tools/mkvariant.sh upf_stamps k_pixel.hip -DLG_UPF_STAMPS): phase stamps of k_upfuse inside one LGT forward, means over the
workgroups.   LGTEUN_HIP_LIB=$PWD/build_variants/upf_stamps.so python tools/upf_stamps.py [C]"""
import ctypes
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = make_module(C, 1)
ops = Ops(net, 128, 128)
z = torch.from_numpy(np.random.default_rng(0).random((32, C, 128, 128)).astype(np.float32)).cuda()
for _ in range(3):
    ops.lgt(0, z)
torch.cuda.synchronize()
n = 4096 * 4 * 12
buf = (ctypes.c_ulonglong * n)()
L = ops.lib
L.lg_debug_upf_stamps.restype = ctypes.c_int
assert L.lg_debug_upf_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4, 12).astype(np.int64)
nwg = 32 * 4 * 16
st = st[:nwg]
names = ['source + parameter loads -> LDS, barrier', 'up-conv on the 160 source pixels, barrier', 'skip loads issued, bicubic taps from LDS', 'skip rows -> exchange buffer, barrier',
         'fusion GEMM, skip half', 'barrier, t rows -> exchange buffer, barrier', 't_save rows out + fusion GEMM, up half', 'barrier, accumulators -> exchange buffer, barrier',
         'output rows out', 'planar LayerNorm half']
print(f'k_upfuse<{4 * C}>, B = 32, 128 x 128: s_memtime ticks, mean over {nwg} workgroups; columns = waves')
for k, nm in enumerate(names):
    d = (st[:, :, k + 1] - st[:, :, k]).mean(axis=0)
    print('  ' + nm.ljust(58), *[f'{v:9.1f}' for v in d])
print('  ' + 'workgroup lifetime'.ljust(58), *[f'{v:9.1f}' for v in (st[:, :, 10] - st[:, :, 0]).mean(axis=0)])
print('  first start -> last end:', int(st[:, :, 10].max() - st[:, :, 0].min()))
