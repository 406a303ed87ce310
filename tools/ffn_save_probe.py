"""GPU probe: the SAVE variant of the fused FFN half-block (forward that keeps the activations for the backward) through lg_op_block_bwd's
forward half is not separable -- this probe times one whole LGT forward with LG_FLAG_SAVE and reports the FFN kernel via lg_prof.
   python tools/ffn_save_probe.py [C] [B] [H]"""
import ctypes
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import make_module
from lgteun_amd import _lib
from lgteun_amd.engine import _ptr, _stream_ptr

C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
net = make_module(C, 1)
eng = net.engine()
plan = eng.plan(H, H)
ws = eng.workspace(plan, B, 1)
z = torch.from_numpy(np.random.default_rng(0).uniform(0, 1, (B, C, H, H)).astype(np.float32)).cuda()
out = torch.empty_like(z)
L = _lib.lib()
_lib.check(L.lg_prof_enable(_lib.KERNEL_IDS['ffn'], 512), 'prof')


def run():
    _lib.check(L.lg_op_lgt(plan, _ptr(eng.flat), 0, _ptr(z), _ptr(out), _ptr(ws), ws.numel(), B, _lib.LG_FLAG_SAVE, 0, _stream_ptr()), 'lgt')


for _ in range(3):
    run()
torch.cuda.synchronize()
L.lg_prof_reset()
for _ in range(5):
    run()
torch.cuda.synchronize()
tot, n = ctypes.c_double(), ctypes.c_int64()
_lib.check(L.lg_prof_read(ctypes.byref(tot), ctypes.byref(n)), 'read')
print(f'LGT forward with SAVE, C={C} B={B} {H}x{H}: fused FFN launches avg {tot.value / n.value * 1e3:.1f} us ({n.value} timed; 4 level-0 + 1 level-1 per LGT)')
