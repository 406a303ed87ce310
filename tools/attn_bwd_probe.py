"""GPU probe: the e = 16 mixer half-block backward alone (lg_op_block_bwd which=1) at the bench geometry: kernel time of k_attn_bwd_f by lg_prof and a
digest of dx + every parameter gradient (two builds whose arithmetic is meant to be the same print the same digest).
   [LGTEUN_HIP_LIB=build_variants/x.so] python tools/attn_bwd_probe.py [reps]"""
import ctypes
import hashlib
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module
from lgteun_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
net = make_module(4, 1)
ops = Ops(net, 128, 128)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
dy = torch.from_numpy(rng.standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
L = _lib.lib()
_lib.check(L.lg_prof_enable(_lib.KERNEL_IDS['attn_bwd'], 4 * reps + 8), 'prof')
for _ in range(2):
    dx, grads = ops.block_bwd(0, 0, 1, x, dy)
torch.cuda.synchronize()
L.lg_prof_reset()
for _ in range(reps):
    dx, grads = ops.block_bwd(0, 0, 1, x, dy)
torch.cuda.synchronize()
tot, n = ctypes.c_double(), ctypes.c_int64()
_lib.check(L.lg_prof_read(ctypes.byref(tot), ctypes.byref(n)), 'read')
L.lg_prof_disable()
h = hashlib.sha256(dx.cpu().numpy().tobytes() + grads.cpu().numpy().tobytes()).hexdigest()[:16]
print(f'attn_bwd: {tot.value / max(n.value, 1) * 1e3:.1f} us per launch ({n.value} timed)  digest {h}  finite {bool(torch.isfinite(dx).all())}  lib={os.environ.get("LGTEUN_HIP_LIB", "default")}')
