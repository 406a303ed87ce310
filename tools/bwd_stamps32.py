import ctypes, os, sys
R = '/root/repo'
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
os.environ['LG_FFN_BWD32'] = 'xs'
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
buf = (ctypes.c_ulonglong * 64)()
f = L.lg_debug_kb_stamps; f.restype = ctypes.c_int
assert f(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(4, 16).astype(np.int64)
names = {1: 'loader: split dh2, LN(x), issue next', 2: 'barrier', 3: 'GEMM phase (4 pixel blocks)', 4: 'barrier', 5: 'LayerNorm backward + dx store'}
prev = st[:, 0].copy()
for i in range(1, 6):
    d = st[:, i] - prev
    print('  ' + names[i].ljust(40), *[str(int(v)).rjust(8) for v in d]); prev = st[:, i].copy()
print('  total', *[str(int(v)).rjust(8) for v in st[:, 5] - st[:, 0]])
