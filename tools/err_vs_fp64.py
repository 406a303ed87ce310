import sys, json, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import load_gold
from helpers import rel_l2
from gpu_helpers import make_module
from oracle import detweights as dw
man=json.load(open('/root/repo/tests/golden/manifest.json'))
for name,m in man.items():
    if not (name.startswith('net_') or (name.startswith('grad_') and 'w' in m)): continue
    g=load_gold(name)
    ms,pan,gt=dw.make_inputs(m['B'],m['C'],m['h'],m.get('w',m['h']),seed=m['seed'],kind=m['kind'])
    net=make_module(m['C'],m['K'])
    with torch.no_grad(): y=net(torch.from_numpy(ms).cuda(),torch.from_numpy(pan).cuda()).cpu().numpy()
    print(f"{name:22s} ours-vs-fp64 {rel_l2(y,g['out_fp64']):.3e}  ref-fp32-vs-fp64 {m['rel_fp32_vs_fp64']:.3e}  ratio {rel_l2(y,g['out_fp64'])/m['rel_fp32_vs_fp64']:.2f}  ours-vs-ref32 {rel_l2(y,g['out_fp32']):.3e}")
