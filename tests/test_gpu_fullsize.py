"""-m gpu: BASELINE.json's full sizes, checked through size-independent properties (the oracle would take minutes here):
data-parallel linearity of the gradient, sample independence of the forward, determinism, faithful == live."""
import numpy as np
import pytest
import torch

from oracle import detweights as dw

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _flat_grad(net, ms, pan, gt):
    from lgteun_amd import FusedAdam
    opt = FusedAdam(net.parameters(), lr=0.0)
    opt.dropout = False
    eng = net.engine()
    loss = float(eng.train_step(ms, pan, gt, opt).item())
    return loss, eng.gflat.clone()


@pytest.mark.parametrize('C,K,B,h', [(4, 4, 32, 32), (8, 4, 32, 32), (8, 8, 16, 64)])      # BASELINE configs[1], configs[2] and configs[4] at its per-GPU batch (16 pairs of 256x256)
def test_gradient_is_linear_in_the_batch(C, K, B, h):
    """mean-L1 gradient of B pairs == mean of the gradients of its two halves: what batch-sharded DDP relies on"""
    from gpu_helpers import make_module
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(B, C, h, h, seed=99, kind='dn'))
    net = make_module(C, K)
    l_all, g_all = _flat_grad(net, ms, pan, gt)
    l_a, g_a = _flat_grad(net, ms[:B // 2], pan[:B // 2], gt[:B // 2])
    l_b, g_b = _flat_grad(net, ms[B // 2:], pan[B // 2:], gt[B // 2:])
    assert abs(l_all - 0.5 * (l_a + l_b)) < 1e-5 * max(1.0, abs(l_all))
    g_half = 0.5 * (g_a + g_b)
    rel = float((g_all - g_half).norm() / g_all.norm())
    assert rel < 2e-4, rel
    assert torch.isfinite(g_all).all() and float(g_all.abs().max()) > 0


def test_full_size_forward_properties():
    from gpu_helpers import make_module
    net = make_module(4, 4)
    ms, pan, _ = (T(a).cuda() for a in dw.make_inputs(32, 4, 32, 32, seed=5, kind='dn'))
    with torch.no_grad():
        net.mode = 'faithful'
        net.faithful_eval = True      # otherwise inference skips the dead stages by itself
        y = net(ms, pan)
        y2 = net(ms, pan)
        net.mode = 'live'
        y_live = net(ms, pan)
        y_tail = net(ms[24:], pan[24:])
    assert torch.equal(y, y2) and torch.equal(y, y_live) and torch.equal(y[24:], y_tail)
    assert torch.isfinite(y).all()


def test_config5_size_batch_independence():
    """BASELINE configs[4] per-GPU shape: C=8, 256x256 PAN, K=8, 2 pairs"""
    from gpu_helpers import make_module
    net = make_module(8, 8)
    ms, pan, _ = (T(a).cuda() for a in dw.make_inputs(2, 8, 64, 64, seed=6, kind='smooth'))
    with torch.no_grad():
        y = net(ms, pan)
        y1 = net(ms[1:], pan[1:])
    assert torch.equal(y[1:], y1) and torch.isfinite(y).all()


def test_dropout_training_mode_statistics():
    """train(): Dropout(0.1) after LGMixer.proj is active (LGT.py:198,215); counter-hash masks differ per call, keep ~90 %,
    and the output stays close to eval() in the mean (statistical parity only)"""
    from gpu_helpers import make_module
    net = make_module(4, 2)
    ms, pan, _ = (T(a).cuda() for a in dw.make_inputs(4, 4, 16, 16, seed=8, kind='smooth'))
    with torch.no_grad():
        y_eval = net(ms, pan)
        net.train()
        ys = torch.stack([net(ms, pan) for _ in range(8)])
        net.eval()
    assert not torch.equal(ys[0], ys[1])
    rel = float((ys.mean(0) - y_eval).norm() / y_eval.norm())
    assert rel < 0.2, rel


@pytest.mark.parametrize('drop', [False, True])
def test_dropout_mask_is_a_function_of_seed_and_consistent_in_backward(drop):
    """The counter-hash dropout mask depends only on (seed, stage, block, element): the same seed reproduces the forward bit
    for bit, another seed changes it, and the backward applies the SAME mask.  The last check is a central-difference
    directional derivative of sum(out * r) with the seed held fixed, along the proj bias of the LAST block: that bias enters
    right behind the mask and everything downstream of it (FFN half-block, tail) is smooth, whereas a direction through any
    FFT mixer is not differentiable numerically (torch.angle's branch cut makes the network piecewise continuous)."""
    from gpu_helpers import make_module
    from lgteun_amd.engine import LG_FLAG_DROPOUT, LG_FLAG_SAVE, LG_FLAG_FAITHFUL
    net = make_module(4, 2)
    eng = net.engine()
    ms, pan, _ = (T(a).cuda() for a in dw.make_inputs(2, 4, 8, 8, seed=21, kind='smooth'))
    flags = (LG_FLAG_DROPOUT if drop else 0) | LG_FLAG_SAVE | LG_FLAG_FAITHFUL
    y1, saved = eng.forward_raw(ms, pan, flags, seed=1234)
    y1 = y1.clone()
    y2, _ = eng.forward_raw(ms, pan, flags, seed=1234)
    assert torch.equal(y1, y2)
    y3, _ = eng.forward_raw(ms, pan, flags, seed=1235)
    assert torch.equal(y1, y3) != drop
    gen = torch.Generator(device='cpu').manual_seed(5)
    r = torch.randn(y1.shape, generator=gen).cuda()
    _, saved = eng.forward_raw(ms, pan, flags, seed=1234)
    g = torch.zeros_like(eng.flat)
    eng.backward_raw(saved, r, g, flags, seed=1234)
    i = [k for k in eng.live_idx if eng.names[k].endswith('fn.fn.proj.bias')][-1]   # last block of the last stage
    o, n = eng.offsets[i], eng.params[i].numel()
    d = torch.zeros_like(eng.flat)
    d[o:o + n] = torch.randn(n, generator=gen).cuda()
    base = eng.flat.clone()
    eps = 1e-2
    vals = []
    for sgn in (+1.0, -1.0):
        eng.flat.copy_(base + sgn * eps * d)
        y, _ = eng.forward_raw(ms, pan, flags, seed=1234)
        vals.append(float((y.double() * r.double()).sum()))
    eng.flat.copy_(base)
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g.double() * d.double()).sum())
    assert abs(fd - an) <= 1e-2 * max(abs(fd), abs(an), 1e-6), (fd, an)


def test_prefetch_loader_stages_batches_on_the_device(tmp_path):
    """lgteun_amd.dataset.PrefetchLoader: batches arrive on the GPU (copied from pinned memory on a side stream, `depth` ahead),
    in order and bit-identical to the host batches; a forward pass consumes them directly"""
    from gpu_helpers import make_module
    from lgteun_amd import dataset as ds
    rng = np.random.default_rng(0)
    root = tmp_path / 'set'
    root.mkdir()
    for i in range(5):
        ds.write_tiff(str(root / f'{i}_lr.tif'), rng.integers(0, 2048, (8, 8, 4)).astype(np.uint16))
        ds.write_tiff(str(root / f'{i}_pan.tif'), rng.integers(0, 2048, (32, 32)).astype(np.uint16))
        ds.write_tiff(str(root / f'{i}_mul.tif'), rng.integers(0, 2048, (32, 32, 4)).astype(np.uint16))
    cfg = dict(dataset=dict(type='PSDataset', image_dirs=[str(root)], bit_depth=11, norm_input=True), batch_size=2, num_workers=0, shuffle=False)
    host, _ = ds.build_loader(cfg)
    dev, _ = ds.build_loader(cfg, device='cuda:0', prefetch_depth=2)
    net = make_module(4, 2)
    n = 0
    for hb, db in zip(host, dev):
        assert db['input_lr'].is_cuda and db['input_pan'].is_cuda and db['target'].is_cuda and db['image_id'] == hb['image_id']
        assert torch.equal(db['input_lr'].cpu(), hb['input_lr']) and torch.equal(db['input_pan'].cpu(), hb['input_pan'])
        with torch.no_grad():
            y = net(db['input_lr'], db['input_pan'])
        assert y.shape == db['target'].shape and torch.isfinite(y).all()
        n += 1
    assert n == 3


def test_training_converges_on_a_fixed_batch():
    """soak: 150 fused train steps (dropout on, default init, lr 1.5e-3) overfit one small batch -- the loss stays finite and
    falls by more than 4x; catches sign / scale errors that per-step parity tests with lr = 0 cannot"""
    import lgteun_amd
    from lgteun_amd.compat import Config
    torch.manual_seed(0)
    net = lgteun_amd.Pansharpening(Config(ms_chans=4), None, stage=4).cuda()
    net.train()
    eng = net.engine()
    opt = lgteun_amd.FusedAdam(net.parameters(), lr=1.5e-3)
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(8, 4, 16, 16, seed=3, kind='smooth'))
    first = float(eng.train_step(ms, pan, gt, opt).item())
    for _ in range(148):
        eng.train_step(ms, pan, gt, opt)
    last = float(eng.train_step(ms, pan, gt, opt).item())
    assert np.isfinite(first) and np.isfinite(last) and last < 0.25 * first, (first, last)


@pytest.mark.parametrize('drop', [False, True])
def test_deferred_dead_stage_forwards_change_nothing(drop):
    """'faithful' training enqueues the K-1 dead-stage LGT forwards on a second stream behind the LGT backward (LG_FLAG_DEFER_DEAD +
    lgteun_dead_forward): the flat gradient and the weights after three Adam steps are bitwise those of the one-stream order"""
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(4, 4, 32, 32, seed=5, kind='dn'))
    res = []
    for overlap in (False, True):
        torch.manual_seed(1234)
        net = make_module(4, 4)
        opt = FusedAdam(net.parameters(), lr=1e-3)
        opt.dropout = drop
        eng = net.engine()
        eng.overlap_dead = overlap
        losses = []
        for _ in range(3):
            losses.append(float(eng.train_step(ms, pan, gt, opt).item()))
        torch.cuda.synchronize()
        res.append((losses, eng.gflat.clone(), eng.flat.clone()))
    assert np.allclose(res[0][0], res[1][0], rtol=1e-6, atol=0)   # the logged loss scalar is summed with float atomics (order varies)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert float(res[1][1].abs().max()) > 0


def _dropout_mask_numpy(seed, stage, blk, first, n):
    """numpy restatement of csrc/common.h mix_seed + dropout_scale (test infrastructure only)"""
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (stage * 8 + blk + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z ^= z >> 31
    idx = np.arange(first, first + n, dtype=np.uint64)
    pair = idx >> np.uint64(1)                 # round 5: one hash per PAIR of elements, 16 bits of it per element
    x = (pair + np.uint64(z & 0xffffffff)).astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= np.uint32(z >> 32) ^ (pair >> np.uint64(32)).astype(np.uint32)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    h = np.where((idx & np.uint64(1)) == 1, x >> np.uint32(16), x & np.uint32(0xffff))
    return np.where(h < np.uint32(6554), np.float32(0), np.float32(1.0 / 0.9))


def test_dropout_mask_rate_independence_and_restatement():
    """lg_dropout_mask exposes the counter-hash mask of the LGMixer dropout: it equals the numpy restatement bit for bit, drops 10 % of
    the elements (within 4 sigma, per seed / stage / block and per channel position), neighbouring elements and neighbouring seeds are
    uncorrelated, and indices beyond 2^32 are well defined"""
    from lgteun_amd import _lib
    from lgteun_amd.engine import _ptr, _stream_ptr
    lib = _lib.lib()
    n = 1 << 20
    out = torch.empty(n, device='cuda')
    masks = {}
    for seed, stage, blk, first in ((1234, 0, 0, 0), (1235, 0, 0, 0), (1234, 3, 4, 0), (2 ** 63 + 77, 1, 2, 0), (1234, 0, 0, (1 << 33) + 5)):
        _lib.check(lib.lg_dropout_mask(seed, stage, blk, first, n, _ptr(out), _stream_ptr()), 'lg_dropout_mask')
        got = out.cpu().numpy()
        np.testing.assert_array_equal(got, _dropout_mask_numpy(seed, stage, blk, first, n))
        drop = got == 0
        sigma = (0.1 * 0.9 / n) ** 0.5
        assert abs(drop.mean() - 0.1) < 4 * sigma, drop.mean()
        per_ch = drop.reshape(-1, 16).mean(0)                      # every channel position of an e = 16 pixel
        assert np.abs(per_ch - 0.1).max() < 4.5 * sigma * 4, per_ch
        assert abs(np.corrcoef(drop[:-1], drop[1:])[0, 1]) < 5e-3      # neighbours
        assert abs(np.corrcoef(drop[:-16], drop[16:])[0, 1]) < 5e-3    # same channel of the next pixel
        masks[(seed, stage, blk, first)] = drop
    keys = list(masks)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            assert abs(np.corrcoef(masks[keys[i]], masks[keys[j]])[0, 1]) < 5e-3, (keys[i], keys[j])
    assert lib.lg_dropout_mask(1, 0, 7, 0, 4, _ptr(out), _stream_ptr()) != 0     # block index out of range
