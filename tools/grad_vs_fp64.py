"""GPU probe: per parameter KIND, this build's L1-gradient distance to the reference's fp64 gradients next to the reference's own fp32
distance (relative L2 over all live tensors of the kind), on the five bench-size / odd-size goldens.  -> profiles/r03_grad_vs_fp64.txt
   python tools/grad_vs_fp64.py"""
import os
import re
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import json
import numpy as np
import torch
from conftest import GOLD, load_gold
from gpu_helpers import make_module
from lgteun_amd import FusedAdam
from oracle import detweights as dw

man = json.load(open(GOLD + '/manifest.json'))
kind_of = lambda k: re.sub(r'^prior_module\.\d+\.((encoder_layers|decoder_layers)\.\d+\.\d+|bottleneck)\.blocks\.\d+\.', 'block.', k)
KINDS = ('global_mixer.conv_amp.0.bias', 'global_mixer.conv_pha.0.bias', 'global_mixer.conv_amp.0.weight', 'global_mixer.conv_pha.0.weight', 'local_mixer.pos_emb')
print('case'.ljust(22), 'kind'.ljust(34), 'ours-vs-fp64'.rjust(13), 'ref32-vs-fp64'.rjust(14), 'ratio'.rjust(7), 'ours-vs-ref32'.rjust(14))
for name in ('grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c4_k2_p80x48', 'grad_c4_k2_p208x176', 'grad_c8_k8_p256'):
    m, g = man[name], load_gold(name)
    g64 = np.load(f'{GOLD}/grad64_{name[5:]}.npz')
    T = torch.from_numpy
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(m['B'], m['C'], m['h'], m['w'], seed=m['seed'], kind=m['kind']))
    net = make_module(m['C'], m['K'])
    opt = FusedAdam(net.parameters(), lr=0.0)
    opt.dropout = False
    eng = net.engine()
    eng.train_step(ms, pan, gt, opt)
    grads = {}
    for i in eng.live_idx:
        n, o, p = eng.names[i], eng.offsets[i], eng.params[i]
        grads[n] = eng.gflat[o:o + p.numel()].view(p.shape).cpu().numpy()
    for kd in KINDS:
        ks = [k for k in grads if k.endswith(kd)]
        t64 = {k: g64['g64/' + k.replace('.', '/')] for k in ks}
        den = sum(float((t64[k] ** 2).sum()) for k in ks) ** 0.5
        ours = sum(float(((grads[k].astype(np.float64) - t64[k]) ** 2).sum()) for k in ks) ** 0.5 / den
        ref = sum(float(((g[k.replace('.', '/')].astype(np.float64) - t64[k]) ** 2).sum()) for k in ks) ** 0.5 / den
        o32 = sum(float(((grads[k].astype(np.float64) - g[k.replace('.', '/')]) ** 2).sum()) for k in ks) ** 0.5 / den
        print(name.ljust(22), kd.ljust(34), f'{ours:13.3e}', f'{ref:14.3e}', f'{ours / max(ref, 1e-30):7.2f}', f'{o32:14.3e}')
