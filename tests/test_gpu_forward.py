"""-m gpu: HIP forward path vs the oracle (same seeded inputs) and vs the reference-generated goldens.
Every call goes through the C ABI (ctypes).  Tolerances: north_star asks 1e-3 relative fp32 for the net;
the fp32 kernels are held far tighter per op."""
import numpy as np
import pytest
import torch

from conftest import load_gold
from helpers import det_params, rel_l2
from oracle import detweights as dw
from oracle import lgteun_oracle as orc

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: -m gpu tests must run on the MI355X box')
    return torch.device('cuda')


def test_library_loads_on_gpu(gpu):
    from lgteun_amd import _lib
    assert b'gfx950' in _lib.lib().lg_version()


@pytest.mark.parametrize('C', [4, 8])
def test_resample_and_data_step(gpu, C):
    from gpu_helpers import Ops, make_module
    net = make_module(C, 2)
    ops = Ops(net, 32, 32)
    g = load_gold(f'ops_c{C}')
    x_ms, z, pan = T(g['resample_in']).cuda(), T(g['z_in']).cuda(), T(g['pan_in']).cuda()
    assert rel_l2(ops.resample(x_ms, 2).cpu(), g['resample_x4']) < 2e-6
    assert rel_l2(ops.resample(x_ms, 1).cpu(), g['resample_x2']) < 2e-6
    assert rel_l2(ops.resample(z, 0).cpu(), g['resample_half']) < 2e-6
    P = det_params(C, 2)
    for stage in (0, 1):
        want = orc.data_step(P, z.cpu(), x_ms.cpu(), pan.cpu(), P[f'eta.{stage}'])
        got = ops.data_step(stage, z, x_ms, pan).cpu()
        assert rel_l2(got, want) < 2e-6
    # stage 0 of a K=1 reference net uses the same shared weights + eta.0 -> golden from the reference itself
    assert rel_l2(ops.data_step(0, z, x_ms, pan).cpu(), g['data_step']) < 2e-6


@pytest.mark.parametrize('C,N,B', [(4, 128, 3), (8, 128, 2), (4, 64, 5), (8, 64, 1)])
def test_one_launch_data_step_is_bitwise_the_tile_kernels(gpu, C, N, B, monkeypatch):
    """k_dstep_fwd (round 4: the whole proximal-gradient step of a (sample, channel) plane in one workgroup's LDS, planes of 128 / 64)
    against the four tile launches it replaces (lg_config.variant LG_VAR_DSTEP_TILES): same arithmetic in the same order -> the update
    AND the three intermediates the backward reads are bit-for-bit equal; and both against the oracle"""
    from gpu_helpers import Ops, make_module
    rng = np.random.default_rng(C * 1000 + N)
    z = T(rng.uniform(0, 1, (B, C, N, N)).astype(np.float32)).cuda()
    ms = T(rng.uniform(0, 1, (B, C, N // 4, N // 4)).astype(np.float32)).cuda()
    pan = T(rng.uniform(0, 1, (B, 1, N, N)).astype(np.float32)).cuda()

    def run():
        ops = Ops(make_module(C, 2), N, N)
        out = torch.empty_like(z)
        tmp = torch.full((3 * z.numel() // 4 + z.numel() // C,), float('nan'), device=z.device)
        from lgteun_amd import _lib
        from lgteun_amd.engine import _ptr, _stream_ptr
        _lib.check(ops.lib.lg_op_data_step(ops.plan, _ptr(ops.eng.flat), 1, _ptr(z), _ptr(ms), _ptr(pan), _ptr(out), _ptr(tmp), B,
                                           _stream_ptr()), 'lg_op_data_step')
        q = z.numel() // 4
        return out, tmp[:q], tmp[q:q + q // 4], tmp[2 * q:3 * q]          # z', t1, r, s1 (api.hip: lg_op_data_step)
    monkeypatch.delenv('LG_DSTEP', raising=False)
    new = run()
    monkeypatch.setenv('LG_DSTEP', 'tiles')
    old = run()
    for a, b, name in zip(new, old, ("z'", 't1', 'r', 's1')):
        assert not torch.isnan(a).any(), name
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    P = det_params(C, 2)
    want = orc.data_step(P, z.cpu(), ms.cpu(), pan.cpu(), P['eta.1'])
    assert rel_l2(new[0].cpu(), want) < 2e-6


@pytest.mark.parametrize('C', [4, 8])
def test_block_pieces(gpu, C):
    """global mixer / mixer half-block / ffn half-block of every block kind vs oracle, on the golden features."""
    from gpu_helpers import Ops, make_module
    net = make_module(C, 1)
    ops = Ops(net, 32, 32)
    g = load_gold(f'ops_c{C}')
    P = det_params(C, 1)
    E = 4 * C
    feat = T(g['feat_in'])                               # [2,32,32,E], sample 1 has negative-DC global planes
    pre = 'prior_module.0.'
    bp = pre + 'encoder_layers.0.0.blocks.0.'
    y = orc.layer_norm(feat, P[bp + '0.fn.norm.weight'], P[bp + '0.fn.norm.bias'])
    want_g = orc.global_mixer(P, bp + '0.fn.fn.global_mixer.', y[..., E // 2:]).permute(0, 3, 1, 2)
    got_g = ops.block(0, 0, 0, feat.cuda()).cpu()
    assert rel_l2(got_g, want_g) < 2e-4
    want_m = feat + orc.lg_mixer(P, bp + '0.fn.fn.', y)
    got_m = ops.block(0, 0, 1, feat.cuda()).cpu()
    assert rel_l2(got_m, want_m) < 1e-4
    y2 = orc.layer_norm(feat, P[bp + '1.fn.norm.weight'], P[bp + '1.fn.norm.bias'])
    want_f = feat + orc.feed_forward(P, bp + '1.fn.fn.', y2)
    got_f = ops.block(0, 0, 2, feat.cuda()).cpu()
    assert rel_l2(got_f, want_f) < 5e-6
    # bottleneck block (level 1: 16x16, 2E channels) on synthetic features
    rng = np.random.default_rng(5)
    f1 = T(rng.standard_normal((2, 16, 16, 2 * E)).astype(np.float32))
    bb = pre + 'bottleneck.blocks.0.'
    y = orc.layer_norm(f1, P[bb + '0.fn.norm.weight'], P[bb + '0.fn.norm.bias'])
    assert rel_l2(ops.block(0, 2, 1, f1.cuda()).cpu(), f1 + orc.lg_mixer(P, bb + '0.fn.fn.', y)) < 1e-4
    y2 = orc.layer_norm(f1, P[bb + '1.fn.norm.weight'], P[bb + '1.fn.norm.bias'])
    assert rel_l2(ops.block(0, 2, 2, f1.cuda()).cpu(), f1 + orc.feed_forward(P, bb + '1.fn.fn.', y2)) < 5e-6


@pytest.mark.parametrize('C', [4, 8])
def test_lgt(gpu, C):
    from gpu_helpers import Ops, make_module
    net = make_module(C, 1)
    ops = Ops(net, 32, 32)
    g = load_gold(f'ops_c{C}')
    got = ops.lgt(0, T(g['z_in']).cuda()).cpu()
    assert rel_l2(got, g['lgt']) < 1e-4                  # vs the reference itself


@pytest.mark.parametrize('name', ['net_c4_k2_p32', 'net_c8_k2_p32', 'net_c4_k4_p64', 'net_c4_k4_p128', 'net_c8_k4_p128',
                                  'net_c4_k2_p256'])
@pytest.mark.parametrize('mode', ['faithful', 'live'])
def test_whole_net_vs_reference_golden(gpu, manifest, name, mode):
    from gpu_helpers import make_module
    m = manifest[name]
    g = load_gold(name)
    ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m['h'], seed=m['seed'], kind=m['kind'])
    net = make_module(m['C'], m['K'])
    net.mode = mode
    net.faithful_eval = True          # run exactly the requested graph (inference skips dead stages by default)
    with torch.no_grad():
        y = net(T(ms).cuda(), T(pan).cuda()).cpu()
    assert y.shape == g['out_fp32'].shape and y.dtype == torch.float32
    # north_star: within 1e-3 relative of the reference PyTorch-CPU fp32 forward
    assert rel_l2(y, g['out_fp32']) < 1e-3
    assert rel_l2(y, g['out_fp64']) < 1e-3
    # PSNR / SAM equal to 3 d.p.
    o = np.transpose(y[0].numpy(), (1, 2, 0)).astype(np.float64) * 2047.5
    t = np.transpose(gt[0], (1, 2, 0)).astype(np.float64) * 2047.5
    met = np.array([orc.psnr(o, t), orc.sam(o, t)])
    assert np.allclose(met, g['metrics'][:2], atol=5e-4), (met, g['metrics'])


def test_config5_shape_c8_k8_p256_vs_oracle(gpu):
    """BASELINE configs[4] shape: 8 bands, 256x256 PAN / 64x64 MS, K=8 (split-FFT path at level 0) vs the oracle"""
    from gpu_helpers import make_module
    C, K = 8, 8
    ms, pan, _ = dw.make_inputs(1, C, 64, 64, seed=5, kind='smooth')
    net = make_module(C, K)
    with torch.no_grad():
        y = net(T(ms).cuda(), T(pan).cuda()).cpu()
        want = orc.forward(det_params(C, K), T(ms), T(pan), K, mode='live')
    assert rel_l2(y, want) < 1e-3


def test_batch_independence_and_determinism(gpu):
    """samples are independent (pure data parallelism) and the path is run-to-run bit-stable"""
    from gpu_helpers import make_module
    net = make_module(4, 2)
    ms, pan, _ = dw.make_inputs(4, 4, 8, 8, seed=3, kind='dn')
    ms, pan = T(ms).cuda(), T(pan).cuda()
    with torch.no_grad():
        y = net(ms, pan)
        y2 = net(ms, pan)
        y_half = net(ms[2:], pan[2:])
    assert torch.equal(y, y2)
    assert torch.equal(y[2:], y_half)


def test_bad_inputs_raise(gpu):
    from gpu_helpers import make_module
    from lgteun_amd._lib import LgteunHipError
    net = make_module(4, 1)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 8, 8, device='cuda'), torch.zeros(1, 1, 32, 32, device='cuda'))
    with pytest.raises(LgteunHipError):      # PAN 24x24: not a multiple of 16
        net(torch.zeros(1, 4, 6, 6, device='cuda'), torch.zeros(1, 1, 24, 24, device='cuda'))
    with pytest.raises(LgteunHipError):      # PAN 1040x16: beyond the 1024-point FFT lines
        net(torch.zeros(1, 4, 260, 4, device='cuda'), torch.zeros(1, 1, 1040, 16, device='cuda'))
    with pytest.raises(RuntimeError):        # no CPU path
        make_module(4, 1, device='cpu')(torch.zeros(1, 4, 8, 8), torch.zeros(1, 1, 32, 32))


@pytest.mark.parametrize('C,K,B,h', [(4, 1, 3, 4), (8, 3, 5, 8), (4, 5, 1, 16)])
def test_edge_shapes_vs_oracle(gpu, C, K, B, h):
    """smallest legal PAN (16x16: one 8x8 window per level-1 plane), odd batch sizes, K = 1 / 3 / 5 (class default)"""
    from gpu_helpers import make_module
    ms, pan, _ = dw.make_inputs(B, C, h, h, seed=40 + B, kind='smooth')
    net = make_module(C, K)
    with torch.no_grad():
        y = net(T(ms).cuda(), T(pan).cuda()).cpu()
        want = orc.forward(det_params(C, K), T(ms), T(pan), K)
    assert rel_l2(y, want) < 1e-4


@pytest.mark.parametrize('impl', ['strip', 'tile', 'xp'])
def test_ffn_forward_ab_kernels_agree_with_the_default(gpu, impl, monkeypatch):
    """the A/B kernels of the fused FFN forward (LG_FFN_IMPL: strip = the exact f32-MFMA strip kernel, in every build; tile = round 1's tile
    kernel and xp = the software-pipelined halo pass, in `make AB=1` builds only -- ADVICE r4: profiles/r05_ab_build_tests.txt is this suite on
    that build) against the default f16-pair kernel on the FFN half-block of both levels (op level: a whole forward would put the FFT mixer's
    branch cut between the two): 2e-6 of the half-block's own contribution"""
    from gpu_helpers import Ops, make_module
    from lgteun_amd._lib import LgteunHipError
    rng = np.random.default_rng(77)
    for blk, e, n in ((0, 16, 32), (2, 32, 16)):
        x = T((rng.standard_normal((2, n, n, e)) * 1.5 + 0.3).astype(np.float32)).cuda()
        monkeypatch.delenv('LG_FFN_IMPL', raising=False)
        want = Ops(make_module(4, 1), 32, 32).block(0, blk, 2, x).double().cpu()
        monkeypatch.setenv('LG_FFN_IMPL', impl)                      # read once per plan: a fresh module builds a fresh plan
        try:
            got = Ops(make_module(4, 1), 32, 32).block(0, blk, 2, x).double().cpu()
        except LgteunHipError as err:
            if 'AB=1' in str(err):
                pytest.skip('this variant is compiled into `make AB=1` builds only')
            raise
        finally:
            monkeypatch.delenv('LG_FFN_IMPL', raising=False)
        den = float((want - x.double().cpu()).norm())
        assert float((got - want).norm()) <= 2e-6 * den, (blk, float((got - want).norm()) / den)


def test_register_chain_ffn_agrees_with_the_channel_split_kernel(gpu, monkeypatch):
    """round 6: k_ffn_xr (a wave owns pixels; LN(x), gelu(h1), gelu(h3) stay in registers) against rounds 2 - 5's k_ffn_xs (LG_FFN_FWD=xs: a wave
    owns hidden channels, operand pieces through LDS) -- the same f16-pair arithmetic under the same scales, products summed in another
    order: 1e-6 of the half-block's own contribution at sizes that exercise partial strips (24 rows), one-tile planes and the bench plane"""
    from gpu_helpers import Ops, make_module
    rng = np.random.default_rng(78)
    for B, n in ((2, 32), (1, 16), (3, 48), (2, 128)):
        x = T((rng.standard_normal((B, n, n, 16)) * 1.5 + 0.3).astype(np.float32)).cuda()
        monkeypatch.delenv('LG_FFN_FWD', raising=False)
        got = Ops(make_module(4, 1), n, n).block(0, 0, 2, x).double().cpu()
        monkeypatch.setenv('LG_FFN_FWD', 'xs')
        try:
            want = Ops(make_module(4, 1), n, n).block(0, 0, 2, x).double().cpu()
        finally:
            monkeypatch.delenv('LG_FFN_FWD', raising=False)
        den = float((want - x.double().cpu()).norm())
        assert float((got - want).norm()) <= 1e-6 * den, (B, n, float((got - want).norm()) / den)


def test_mixer_f16_pair_products_agree_with_the_bf16_triples(gpu, monkeypatch):
    """round 6: to_qkv and Q K^T of k_attn_m multiply f16 pairs under static operand scales (bounds from the block's LayerNorm affine and to_qkv
    weights, k_ffn_prep.hip) instead of bf16 triples (LG_ATTN_SPLIT=bf16x3): the mixer half-block of both levels, C = 4 and 8 (head dimensions
    4, 8, 16), agrees to 2e-6 of its own contribution -- also with the LayerNorm affine and the to_qkv weights pushed out of f16's range in
    opposite directions"""
    from gpu_helpers import Ops, make_module
    rng = np.random.default_rng(79)
    for C in (4, 8):
        for blk, e, n in ((0, 4 * C, 32), (2, 8 * C, 16)):
            for mult in ({}, {'norm.weight': 1e3, 'norm.bias': 1e3, 'fn.local_mixer.to_qkv.weight': 1e-3}, {'norm.weight': 1e-4, 'norm.bias': 1e-4, 'fn.local_mixer.to_qkv.weight': 3e3}):
                x = T((rng.standard_normal((2, n, n, e)) * 1.5 + 0.3).astype(np.float32)).cuda()
                got = {}
                for split in ('f16x2', 'bf16x3'):
                    monkeypatch.setenv('LG_ATTN_SPLIT', split)
                    net = make_module(C, 1)
                    sd = net.state_dict()
                    pre = 'prior_module.0.' + ('encoder_layers.0.0.blocks.0.' if blk == 0 else 'bottleneck.blocks.0.') + '0.fn.'
                    for k, f in mult.items():
                        sd[pre + k] = sd[pre + k] * f
                    net.load_state_dict(sd)
                    got[split] = Ops(net, 32, 32).block(0, blk, 1, x).double().cpu()
                monkeypatch.delenv('LG_ATTN_SPLIT', raising=False)
                den = float((got['bf16x3'] - x.double().cpu()).norm())
                err = float((got['f16x2'] - got['bf16x3']).norm()) / den
                assert torch.isfinite(got['f16x2']).all() and err <= 2e-6, (C, blk, sorted(mult), err)
