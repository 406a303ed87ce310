"""GPU probe: the window-attention kernels alone at the bench geometry, timed with the library's HIP-event facility.
   python tools/attn_probe.py [fwd|bwd] [C] [blk] [B] [H]"""
import ctypes
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module
from lgteun_amd import _lib

what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
H = int(sys.argv[5]) if len(sys.argv) > 5 else 128
net = make_module(C, 1)
ops = Ops(net, H, H)
e = 4 * C * (2 if blk == 2 else 1)
h = H // 2 if blk == 2 else H
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, h, h, e)).astype(np.float32)).cuda()
dy = torch.from_numpy(rng.standard_normal((B, h, h, e)).astype(np.float32)).cuda()
L = _lib.lib()
kid = _lib.KERNEL_IDS['attn' if what == 'fwd' else 'attn_bwd']
_lib.check(L.lg_prof_enable(kid, 256), 'prof')
run = (lambda: ops.block(0, blk, 1, x)) if what == 'fwd' else (lambda: ops.block_bwd(0, blk, 1, x, dy))
for _ in range(3):
    y = run()
torch.cuda.synchronize()
L.lg_prof_reset()
for _ in range(10):
    y = run()
torch.cuda.synchronize()
tot, n = ctypes.c_double(), ctypes.c_int64()
_lib.check(L.lg_prof_read(ctypes.byref(tot), ctypes.byref(n)), 'read')
out = y if what == 'fwd' else y[0]
print(f'attention {what} C={C} blk={blk} e={e} B={B} {h}x{h}: {tot.value / n.value * 1e3:.1f} us per launch group ({n.value} timed)  checksum {float(out.double().sum()):.6f}')
