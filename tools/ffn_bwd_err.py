"""GPU diagnostic: FFN half-block backward (k_ffn_dw_bwd_xs + k_ffn1_bwd_xs) against fp64 autograd over the oracle, per tensor, for the
f16-pair (default) and the bf16-triple (LG_FFN_BWD_SPLIT=bf16x3) products of the pixelwise half.      python tools/ffn_bwd_err.py [gscale]"""
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np
import torch
from gpu_helpers import Ops, make_module
from test_gpu_backward import _oracle_block

gscale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
BIG = len(sys.argv) > 2 and sys.argv[2] == 'big'      # 4 x 128 x 128 pixels at e = 16 (65 536-term pixel sums) instead of 2 x 32 x 32
T = torch.from_numpy
rng = np.random.default_rng(23)
for blk, e, n in (((0, 16, 128),) if BIG else ((0, 16, 32), (2, 32, 16))):
    Bn, HW = (4, 128) if BIG else (2, 32)
    pre = 'prior_module.0.' + ('encoder_layers.0.0.blocks.0.' if blk == 0 else 'bottleneck.blocks.0.') + '1.fn.'
    x = T((rng.standard_normal((Bn, n, n, e)) * 1.5 + 0.3).astype(np.float32))
    dy = T((rng.standard_normal((Bn, n, n, e)) * gscale).astype(np.float32))
    got = {}
    for split in ('f16x2', 'bf16x3'):
        os.environ['LG_FFN_BWD_SPLIT'] = split
        net = make_module(4, 1)
        ops = Ops(net, HW, HW)
        dx, grads = ops.block_bwd(0, blk, 2, x.cuda(), dy.cuda())
        names = [nm for nm in ops.eng.names if nm.startswith(pre)]
        got[split] = {'dx': dx.double().cpu(), **{nm: ops.grad_of(grads, nm).double().cpu() for nm in names}}
    P64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
    want_dx, want_g = _oracle_block(P64, 4, blk, 2, x.double(), dy.double())
    want = {'dx': want_dx.detach(), **{k: v for k, v in want_g.items() if k.startswith(pre)}}
    print(f'block {blk} (e = {e}), upstream gradient x {gscale:g}:   relative L2 error against fp64    f16 pairs    bf16 x 3')
    for k, ref in want.items():
        rn = float(ref.norm())
        e2, e3 = (float((got[sp][k].reshape(ref.shape) - ref).norm()) / rn for sp in ('f16x2', 'bf16x3'))
        print(f'  {k[len(pre):] if k != "dx" else k:32s} {e2:12.3e} {e3:12.3e}')
