import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import torch
from gpu_helpers import make_module
from oracle import detweights as dw
net = make_module(4, 4)
B = 32
ms, pan, gt = dw.make_inputs(B, 4, 32, 32, seed=1, kind='dn')
ms, pan = torch.from_numpy(ms).cuda(), torch.from_numpy(pan).cuda()
for mode in ('faithful', 'live'):
    net.mode = mode
    with torch.no_grad():
        for _ in range(3): net(ms, pan)
        torch.cuda.synchronize()
        t = time.time()
        n = 10
        for _ in range(n): net(ms, pan)
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
    print(mode, 'fwd ms', dt * 1e3, 'pairs/s', B / dt)
