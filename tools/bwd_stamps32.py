import ctypes, os, sys
R = '/root/repo'
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
if len(sys.argv) < 2 or sys.argv[1] == 'kb':
    os.environ.pop('LG_FFN_BWD32', None)   # k_ffn1_bwd_xs<32> is the default since round 5
import numpy as np, torch
from gpu_helpers import Ops, make_module
net = make_module(8, 1)
ops = Ops(net, 128, 128)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((32, 128, 128, 32)).astype(np.float32)).cuda()
dy = torch.from_numpy(rng.standard_normal((32, 128, 128, 32)).astype(np.float32)).cuda()
for _ in range(3):
    ops.block_bwd(0, 0, 2, x, dy)
torch.cuda.synchronize()
L = ops.lib
import sys as _s
which = _s.argv[1] if len(_s.argv) > 1 else 'kb'
buf = (ctypes.c_ulonglong * (64 if which == 'kb' else 128))()
f = getattr(L, 'lg_debug_%s_stamps' % which); f.restype = ctypes.c_int
assert f(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
names = {1: 'loader: split dh2, LN(x), issue next', 2: 'barrier', 3: 'GEMM phase (4 pixel blocks)', 4: 'barrier', 5: 'LayerNorm backward + dx store'}
if which == 'ka':
    names = {1: 'h3 fetch c0 + dy store c0', 2: 'barrier', 3: 'chunk 0', 4: 'barrier', 5: 'chunk 1', 6: 'barrier', 7: 'chunk 2 (+ h2 requests)', 8: 'barrier',
             9: 'taps reload', 10: 'P2: items (dw^T, tap gradients, dh2 store)'}
prev = st[:, 0].copy()
for i in range(1, 6 if which == 'kb' else 11):
    d = st[:, i] - prev
    print('  ' + names[i].ljust(40), *[str(int(v)).rjust(8) for v in d]); prev = st[:, i].copy()
print('  total', *[str(int(v)).rjust(8) for v in st[:, 5 if which == 'kb' else 10] - st[:, 0]])
