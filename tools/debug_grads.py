import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import json, numpy as np, torch
from gpu_helpers import make_module
from helpers import det_params
from oracle import detweights as dw, lgteun_oracle as orc
name = sys.argv[1] if len(sys.argv) > 1 else 'grad_c4_k2_p32'
m = json.load(open(R + '/tests/golden/manifest.json'))[name]
g = np.load(R + f'/tests/golden/{name}.npz')
ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m['h'], seed=m['seed'], kind=m['kind'])
T = torch.from_numpy
net = make_module(m['C'], m['K'])
out = net(T(ms).cuda(), T(pan).cuda())
loss = torch.nn.functional.l1_loss(out, T(gt).cuda()); loss.backward()
P64 = det_params(m['C'], m['K'], dtype=torch.float64, requires_grad=True)
l64 = orc.l1_loss(orc.forward(P64, T(ms).double(), T(pan).double(), m['K']), T(gt).double()); l64.backward()
rows = []
for k, p in net.named_parameters():
    if p.grad is None: continue
    ref = g[k.replace('.', '/')]; r64 = P64[k].grad.numpy()
    sc = max(np.abs(r64).max(), 1e-9)
    rows.append((np.abs(p.grad.cpu().numpy() - r64).max() / sc, np.abs(ref - r64).max() / sc, k.replace('prior_module.1.', ''), float(sc)))
rows.sort(reverse=True)
print('loss', loss.item(), float(g['loss']), l64.item())
print('%-10s %-10s %s' % ('hip-vs-64', 'ref32-vs-64', 'name'))
for r in rows[:45]: print('%.3e  %.3e  %s  (max %.2e)' % r)
