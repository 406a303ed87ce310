"""GPU probe: WHERE the distance of this build's L1 gradients to the reference's fp64 gradients sits in the cases where it
is larger than the reference's own fp32 distance -- per block and per channel of the cancelling-sum kinds, under the arithmetic switches
that separate the candidates (split-bf16 FFN GEMMs vs the exact f32-MFMA kernels).
   python tools/grad_noise_probe.py [case ...]      -> text for profiles/r04_grad_vs_fp64.txt"""
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
KINDS = ('global_mixer.conv_amp.0.bias', 'global_mixer.conv_pha.0.bias', 'global_mixer.conv_amp.0.weight', 'global_mixer.conv_pha.0.weight',
         'local_mixer.pos_emb')
BLK = re.compile(r'^prior_module\.\d+\.((?:encoder_layers|decoder_layers)\.\d+\.\d+|bottleneck)\.blocks\.(\d+)\.')


def grads_of(name, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = man[name]
        T = torch.from_numpy
        ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(m['B'], m['C'], m['h'], m.get('w', m['h']), seed=m['seed'], kind=m['kind']))
        net = make_module(m['C'], m['K'])
        opt = FusedAdam(net.parameters(), lr=0.0)
        opt.dropout = False
        eng = net.engine()
        eng.train_step(ms, pan, gt, opt)
        out = {}
        for i in eng.live_idx:
            n, o, p = eng.names[i], eng.offsets[i], eng.params[i]
            out[n] = eng.gflat[o:o + p.numel()].view(p.shape).cpu().numpy().astype(np.float64)
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    cases = sys.argv[1:] or ['grad_c8_k4_p128', 'grad_c4_k2_p208x176']
    switches = [('default', {}), ('LG_FFN_IMPL=strip', {'LG_FFN_IMPL': 'strip'})]
    for name in cases:
        g32 = load_gold(name)
        g64 = np.load(f'{GOLD}/grad64_{name[5:]}.npz')
        runs = [(lbl, grads_of(name, env)) for lbl, env in switches]
        print(f'== {name}: per block, relative L2 to the reference fp64 gradient  (ref32 = the reference\'s own fp32 run)')
        print('block'.ljust(34), 'kind'.ljust(32), 'ref32'.rjust(10), *[lbl.rjust(18) for lbl, _ in runs])
        for k in sorted(runs[0][1]):
            kd = [x for x in KINDS if k.endswith(x)]
            if not kd:
                continue
            t64 = g64['g64/' + k.replace('.', '/')]
            den = float((t64 ** 2).sum()) ** 0.5
            ref = float(((g32[k.replace('.', '/')].astype(np.float64) - t64) ** 2).sum()) ** 0.5 / den
            mm = BLK.match(k)
            print((mm.group(1) + '.' + mm.group(2)).ljust(34), kd[0].split('.', 1)[1].ljust(32), f'{ref:10.2e}',
                  *[f'{float(((g[k] - t64) ** 2).sum()) ** 0.5 / den:18.2e}' for _, g in runs])
        # per channel of the phase weights: a single flipped angle() bin shows up as ONE channel of ONE block
        print(f'-- {name}: conv_pha.0.weight per channel, |ours - fp64| / max|fp64| of the tensor (default switch), reference fp32 beside it')
        for k in sorted(runs[0][1]):
            if not k.endswith('global_mixer.conv_pha.0.weight'):
                continue
            t64 = g64['g64/' + k.replace('.', '/')].ravel()
            ours = runs[0][1][k].ravel()
            ref = g32[k.replace('.', '/')].astype(np.float64).ravel()
            sc = np.abs(t64).max()
            mm = BLK.match(k)
            print((mm.group(1) + '.' + mm.group(2)).ljust(34), 'ours', ' '.join(f'{abs(a - b) / sc:8.1e}' for a, b in zip(ours, t64)))
            print(' ' * 34, 'ref ', ' '.join(f'{abs(a - b) / sc:8.1e}' for a, b in zip(ref, t64)))


if __name__ == '__main__':
    main()
