"""GPU probe (diagnostic build, tools/build_stamps.sh): phase stamps of the fused local-mixer backward k_attn_bwd_f, workgroup 0, per wave.
   LGTEUN_HIP_LIB=$PWD/lgteun_amd/_lgteun_hip_stamps.so python tools/attn_bwd_stamps.py"""
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
    ops.block_bwd(0, 0, 1, x, dy)
torch.cuda.synchronize()
L = ops.lib
from lgteun_amd._lib import KERNEL_IDS
L.lg_prof_enable(KERNEL_IDS['attn_bwd'], 64)
ops.block_bwd(0, 0, 1, x, dy)
tot, n = ctypes.c_double(), ctypes.c_int64()
L.lg_prof_read(ctypes.byref(tot), ctypes.byref(n))
L.lg_prof_disable()
print(f'k_attn_bwd_f by HIP events (this build): {1e3 * tot.value / max(n.value, 1):.1f} us per launch, {n.value} launch(es)')
N = 512 * 8 * 8 * 16
buf = (ctypes.c_ulonglong * N)()
f = L.lg_debug_kf_stamps
f.restype = ctypes.c_int
assert f(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 8, 16).astype(np.int64)   # [workgroup][wave][group][stamp]
st = st[st[:, 0, 0, 11] > 0][:, :(4 if st[0, 4, 0, 11] == 0 else 8)]             # the workgroups / waves that ran
names = {1: 'prologue (LN, qkv, dO -> tiles)', 2: 'barrier B1', 3: 'pass 1 (lane = query)', 4: 'epilogue loads issued + pass 2 (lane = key)',
         5: 'next loads issued + E1: to_qkv^T partial, image', 6: 'to_qkv weight-gradient MFMAs', 7: 'barrier B2', 8: 'E3: LN backward, dx, slots, images',
         9: 'proj weight-gradient MFMAs', 10: 'barrier B3'}
print('k_attn_bwd_f, s_memtime ticks (~2.3 GHz) per phase, mean over the workgroups; columns = waves (window slot, head)')
for it in (0, 3, 7):
    print(f' window group {it} of the workgroup')
    for i in range(1, 11):
        d = (st[:, :, it, i] - st[:, :, it, i - 1]).mean(axis=0)
        print('  ' + names[i].ljust(50), *[str(int(v)).rjust(7) for v in d])
    print('  ' + 'total'.ljust(50), *[str(int(v)).rjust(7) for v in (st[:, :, it, 10] - st[:, :, it, 0]).mean(axis=0)])
print('  ' + 'staging (kernel start -> first group)'.ljust(50), *[str(int(v)).rjust(7) for v in (st[:, :, 0, 0] - st[:, :, 0, 11]).mean(axis=0)])
print('  ' + 'whole loop (8 groups)'.ljust(50), *[str(int(v)).rjust(7) for v in (st[:, :, 0, 12] - st[:, :, 0, 0]).mean(axis=0)])
print('  ' + 'write-out'.ljust(50), *[str(int(v)).rjust(7) for v in (st[:, :, 0, 13] - st[:, :, 0, 12]).mean(axis=0)])
print('  kernel start spread over workgroups (ticks):', int(st[:, 0, 0, 11].max() - st[:, 0, 0, 11].min()), ' kernel end spread:', int(st[:, 0, 0, 13].max() - st[:, 0, 0, 13].min()), ' first start -> last end:', int(st[:, 0, 0, 13].max() - st[:, 0, 0, 11].min()))
