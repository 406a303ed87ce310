"""GPU diagnostic: whole-net forward error against the reference's fp64 result, per golden and per kernel choice
(the matrix-pipe / vector-pipe local mixer x the split 16-bit / exact-f32-MFMA FFN x the real-input / complex-row FFT mixer).  The FFT mixer's angle() branch cut turns a 1e-7
perturbation of a near-negative-real bin into a 1e-5 .. 1e-4 output change, so on some inputs two equally accurate fp32 evaluations
differ from fp64 by amounts 1000x apart: the table shows which goldens have such a bin.     python tools/err_vs_fp64.py"""
import json
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from conftest import load_gold
from helpers import rel_l2
from gpu_helpers import make_module
from oracle import detweights as dw

man = json.load(open(R + '/tests/golden/manifest.json'))
combos = [('m+split', {}), ('m+strip', {'LG_FFN_IMPL': 'strip'}), ('valu+split', {'LG_ATTN_FWD': 'valu'}), ('valu+strip', {'LG_ATTN_FWD': 'valu', 'LG_FFN_IMPL': 'strip'})]
# x the FFT mixer kernel: r = real-input rows (k_fftmix_r, the default), f = complex rows (LG_FFT=full)
combos = [(n + '+' + f, dict(e, **({'LG_FFT': 'full'} if f == 'f' else {}))) for f in 'rf' for n, e in combos]
print(f"{'golden':22s} " + ' '.join(f'{c[0]:>12s}' for c in combos) + '   ref fp32-vs-fp64')
for name, m in man.items():
    if not (name.startswith('net_') or (name.startswith('grad_') and 'w' in m)):
        continue
    g = load_gold(name)
    ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m.get('w', m['h']), seed=m['seed'], kind=m['kind'])
    errs = []
    for _, env in combos:
        for k in ('LG_FFN_IMPL', 'LG_ATTN_FWD', 'LG_FFT'):
            os.environ.pop(k, None)
        os.environ.update(env)
        net = make_module(m['C'], m['K'])
        with torch.no_grad():
            y = net(torch.from_numpy(ms).cuda(), torch.from_numpy(pan).cuda()).cpu().numpy()
        errs.append(rel_l2(y, g['out_fp64']))
    print(f"{name:22s} " + ' '.join(f'{e:12.3e}' for e in errs) + f"   {m['rel_fp32_vs_fp64']:.3e}")
