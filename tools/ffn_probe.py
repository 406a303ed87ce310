"""GPU probe: the fused feed_forward half-block alone (lg_op_block which=2) at the bench geometry -- time per launch and, under
rocprofv3 --pmc, its counters.   python tools/ffn_probe.py [C] [blk] [B] [H] [reps]"""
import os
import sys
import time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
H = int(sys.argv[4]) if len(sys.argv) > 4 else 128
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
net = make_module(C, 1)
ops = Ops(net, H, H)
e = 4 * C * (2 if blk == 2 else 1)
h = H // 2 if blk == 2 else H
x = torch.from_numpy(np.random.default_rng(0).standard_normal((B, h, h, e)).astype(np.float32)).cuda()
for _ in range(3):
    y = ops.block(0, blk, 2, x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for i in range(reps):
    y = ops.block(0, blk, 2, x)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
# the FFN kernel alone: the library's live timing facility (HIP events around the launch, on its stream) -- the op entry also runs the scale prep
import ctypes
from lgteun_amd import _lib
L = _lib.lib()
_lib.check(L.lg_prof_enable(_lib.KERNEL_IDS['ffn'], 4 * reps), 'lg_prof_enable')
for i in range(reps):
    y = ops.block(0, blk, 2, x)
torch.cuda.synchronize()
tot_ms, n_l = ctypes.c_double(), ctypes.c_int64()
_lib.check(L.lg_prof_read(ctypes.byref(tot_ms), ctypes.byref(n_l)), 'lg_prof_read')
L.lg_prof_disable()
print(f'kernel alone (lg_prof, {n_l.value} launches): {tot_ms.value / max(n_l.value, 1) * 1e3:.2f} us   lib={os.environ.get("LGTEUN_HIP_LIB", "default")} LG_FFN_FWD={os.environ.get("LG_FFN_FWD", "")}')
px = B * h * h
flops = (2 * (e * 4 * e + 4 * e * 4 * e + 4 * e * e) + 18 * 4 * e) * px
print(f'ffn half-block C={C} blk={blk} e={e} B={B} {h}x{h}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us  '
      f'-> {flops / ts[len(ts) // 2] / 1e6:.1f} TFLOP/s algorithmic  impl={os.environ.get("LG_FFN_IMPL", "split")}  checksum {float(y.double().sum()):.6f}')
