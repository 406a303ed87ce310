"""-m gpu: train-step parity at the sizes the headline is measured on, against gradients the REFERENCE ITSELF produced
(tests/golden/grad_*_p128 / _p256 / non-pow2; tools/gen_goldens.py round2): the bench size (128x128 PAN, K=4, C=4 and C=8 --
k_ffn_strip<SAVE>, the e=32 fused FFN and the e=64 pair with their backwards), BASELINE configs[4]'s shape (C=8, 256x256 PAN,
K=8: split-FFT forward AND backward) and two PAN sizes that are not powers of two (Bluestein mixer; pinned directly against the
reference, not through the oracle).  Plus the two-call backward the data-parallel step uses.  Dropout off."""
import numpy as np
import pytest
import torch

from conftest import GOLD, load_gold
from helpers import rel_l2
from oracle import detweights as dw

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _live_grads(eng):
    out = {}
    for i in eng.live_idx:
        n, o, p = eng.names[i], eng.offsets[i], eng.params[i]
        out[n] = eng.gflat[o:o + p.numel()].view(p.shape).cpu().numpy()
    return out


@pytest.mark.parametrize('name', ['grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c4_k2_p80x48', 'grad_c4_k2_p208x176', 'grad_c8_k8_p256'])
@pytest.mark.parametrize('mode', ['faithful', 'live'])
def test_train_step_vs_reference_gradients(manifest, name, mode):
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    m, g = manifest[name], load_gold(name)
    if mode == 'live' and name != 'grad_c4_k4_p128':
        pytest.skip('live == faithful is bitwise (test_gpu_fullsize); one size is enough here')
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(m['B'], m['C'], m['h'], m['w'], seed=m['seed'], kind=m['kind']))
    net = make_module(m['C'], m['K'])
    net.mode = mode
    net.faithful_eval = True
    with torch.no_grad():
        y = net(ms, pan).cpu().numpy()
    # forward: north_star's 1e-3 against the reference's fp32 AND fp64 outputs
    assert rel_l2(y, g['out_fp32']) < 1e-3 and rel_l2(y, g['out_fp64']) < 1e-3, (rel_l2(y, g['out_fp32']), rel_l2(y, g['out_fp64']))
    opt = FusedAdam(net.parameters(), lr=0.0)
    opt.dropout = False
    eng = net.engine()
    loss = float(eng.train_step(ms, pan, gt, opt).item())
    assert abs(loss - float(g['loss'])) < 2e-5 * max(1.0, float(g['loss']))
    grads = _live_grads(eng)
    assert len(grads) == len(g.files) - 5                     # loss, loss_fp64, out_fp32, out_fp64, self_err + one entry per live tensor
    dead = [n for n in eng.names if n.startswith(tuple(f'prior_module.{i}.' for i in m['none_grad_stages']))]
    assert len(dead) == m['n_none'] and not set(dead) & set(grads)
    num = sum(float(((v.astype(np.float64) - g[k.replace('.', '/')]) ** 2).sum()) for k, v in grads.items())
    den = sum(float((g[k.replace('.', '/')].astype(np.float64) ** 2).sum()) for k in grads)
    err = (num / den) ** 0.5
    # gate: global relative L2 1e-3; for scale, the reference's own fp32 gradients are m['grad_rel_fp32_vs_fp64'] off its fp64 ones
    assert err < 1e-3, (err, m['grad_rel_fp32_vs_fp64'])
    # per tensor, on the tensor's own scale (floored at 2e-5): 3e-2 against the reference's fp32 gradient -- for every tensor except the
    # five KINDS whose L1 gradients are sums that cancel to ~1e-5 of their terms (the FFT mixer's amplitude / phase scale + bias, pos_emb).
    # An fp32 golden cannot judge those: tests/golden/gradnoise.json (tools/gen_goldens.py --only-r4, the reference itself) records that
    # the reference's OWN fp32 gradients of these kinds move by 0.1 % ... 9 % when its inputs are nudged by one ulp -- the network is
    # piecewise continuous (torch.angle's branch cut, abs() behind irfft2), so one fp32 evaluation lands anywhere in that band around
    # the fp64 value, and a case where the reference's fp32 happens to sit 10 x closer (grad_c8_k4_p128) is a lucky draw
    # (profiles/r04_grad_vs_fp64.txt).  The truth is the reference's fp64 gradient (grad64_*.npz); the gate is PER CASE (VERDICT r3
    # item 3): per kind (relative L2 over the kind's live tensors) this build is no further from fp64 than 3 x the larger of
    # (a) the reference's own fp32 distance in THIS case and (b) the reference's own largest one-ulp spread in THIS case; per tensor of
    # these kinds, on the tensor's own scale, 5 x the same two numbers of that tensor.
    import json
    kinds = ('global_mixer.conv_amp.0.bias', 'global_mixer.conv_pha.0.bias', 'global_mixer.conv_amp.0.weight', 'global_mixer.conv_pha.0.weight',
             'local_mixer.pos_emb')
    with open(f'{GOLD}/gradnoise.json') as f:
        noise = json.load(f)[name]
    g64c = np.load(f'{GOLD}/grad64_{name[5:]}.npz')
    report = {}
    for kd in kinds:
        ks = [k for k in grads if k.endswith(kd)]
        t64 = {k: g64c['g64/' + k.replace('.', '/')] for k in ks}
        den = sum(float((t64[k] ** 2).sum()) for k in ks) ** 0.5
        ours = sum(float(((grads[k].astype(np.float64) - t64[k]) ** 2).sum()) for k in ks) ** 0.5 / den
        nk = noise[kd]
        bound = min(3.0 * max(nk['ref_vs_fp64'], max(nk['ref_spread'])), 0.1)   # capped: the band-against-band test below is the discriminating gate
        report[kd] = (ours, nk['ref_vs_fp64'], max(nk['ref_spread']))
        assert ours <= bound, (name, kd, ours, nk)
        for k in ks:
            nt = noise['tensors'][k]
            e_t = float(((grads[k].astype(np.float64) - t64[k]) ** 2).sum()) ** 0.5 / float((t64[k] ** 2).sum()) ** 0.5
            assert e_t <= min(5.0 * max(nt['ref_vs_fp64'], max(nt['ref_spread'])), 0.25), (name, k, e_t, nt)   # (kernel-level truth: test_mixer_backward_kernel_at_awkward_shapes, 1e-4 vs fp64)
    print(name, mode, {k: tuple(f'{v:.2e}' for v in r) for k, r in report.items()})
    worst = max((float(np.abs(v - g[k.replace('.', '/')]).max() / max(float(np.abs(g[k.replace('.', '/')]).max()), 2e-5)) / 3e-2, k)
                for k, v in grads.items() if not k.endswith(kinds))
    assert worst[0] < 1.0, worst
    a, b = eng.live_ranges[0][1], eng.live_ranges[1][0]
    assert float(eng.gflat[a:b].abs().max()) == 0.0            # dead-stage slots of the flat gradient buffer: never written


@pytest.mark.parametrize('name', ['grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c8_k8_p256'])
def test_bf16_mode_at_the_measured_sizes(manifest, name):
    """precision='bf16' -- the mode BASELINE configs[1] names and bench.py's `bf16_mode` measures -- gated WHERE it is measured (VERDICT r5
    item 2): configs[1]'s shape (C = 4, 128 x 128, K = 4), configs[2]'s (C = 8) and configs[4]'s single-GPU shape (C = 8, 256 x 256, K = 8), i.e.
    the NP = 1 instances of every width (e = 16 / 32 / 64), the tanh-GELU pair and bf16 storage of the saved tensors, against the
    REFERENCE's own outputs and gradients: forward within 5e-3 relative and >= 50 dB PSNR of the default mode AND within 1e-2 of the
    reference's fp64 output; loss 2e-3; global gradient relative L2 2e-2 against the reference's fp32 gradients."""
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    m, g = manifest[name], load_gold(name)
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(m['B'], m['C'], m['h'], m['w'], seed=m['seed'], kind=m['kind']))
    net = make_module(m['C'], m['K'])
    net.faithful_eval = True
    with torch.no_grad():
        y32 = net(ms, pan)
    net.precision = 'bf16'
    with torch.no_grad():
        y16 = net(ms, pan)
    fwd_rel = rel_l2(y16.cpu(), y32.cpu())
    psnr = 10 * np.log10(1.0 / max(float(((y16 - y32) ** 2).mean()), 1e-30))
    ref_rel = rel_l2(y16.cpu().numpy(), g['out_fp64'])
    assert fwd_rel < 5e-3 and psnr >= 50.0 and ref_rel < 1e-2, (fwd_rel, psnr, ref_rel)
    opt = FusedAdam(net.parameters(), lr=0.0)
    opt.dropout = False
    eng = net.engine()
    loss = float(eng.train_step(ms, pan, gt, opt).item())
    assert abs(loss - float(g['loss'])) < 2e-3 * abs(float(g['loss'])), (loss, float(g['loss']))
    grads = _live_grads(eng)
    num = sum(float(((v.astype(np.float64) - g[k.replace('.', '/')]) ** 2).sum()) for k, v in grads.items())
    den = sum(float((g[k.replace('.', '/')].astype(np.float64) ** 2).sum()) for k in grads)
    err = (num / den) ** 0.5
    print(f'bf16 mode {name}: forward {fwd_rel:.2e} of the default mode ({psnr:.1f} dB), {ref_rel:.2e} of the reference fp64; loss {loss:.6f} / {float(g["loss"]):.6f}; gradient {err:.2e}')
    assert err < 2e-2, err


@pytest.mark.parametrize('name', ['grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c4_k2_p80x48', 'grad_c4_k2_p208x176', 'grad_c8_k8_p256'])
def test_cancelling_sum_gradient_kinds_band_against_band(manifest, name):
    """The five gradient kinds that cancel to ~1e-5 of their terms, gated BAND AGAINST BAND: this build's gradients go
    through the same four seeded +-1-ulp input nudges the reference's own fp32 gradients went through (tools/gen_goldens.py round4 ->
    tests/golden/gradnoise.json; tools/grad_spread.py is the function and the table, profiles/r05_grad_spread.txt), and per kind
      (i)  the rms of this build's spread is at most 2 x the reference's (measured 0.0 ... 1.4 x): the build's band is no wider;
      (ii) the MEAN of the five draws is within 2 x max(the reference's one-run distance, the reference's spread rms) of the fp64 gradient
           (measured 0.1 ... 0.65 x on the power-of-two sizes): the band is centred where the reference's is.  3 x on the two non-power-of-two
           cases, whose FFT mixer runs Bluestein lines: their offset (up to 2.8 x at 208 x 176) is deterministic fp32 arithmetic of another
           algorithm (pocketfft's mixed radix on the reference's side), which nudged inputs do not average out.
    A one-draw comparison cannot tell "inside the band" from "a 3 x wider band"; this can.  What it can and cannot see, checked with
    k_attn_bwd_f's pos_emb accumulation degraded on purpose (`-DLG_DEGRADE_DPOS` variant builds, LGTEUN_HIP_LIB): a 5 % error in what
    enters the pos_emb gradient turns (ii) red at 208 x 176 (4.65 against its limit 3; 2.76 without) and moves grad_c4_k4_p128 from 0.50
    to 1.9 of its limit 2; rounding every dS to bf16 before it is added changes nothing measurable (0.93 -> 0.94): random 2^-9 errors
    of 16 k terms average out far below the band the reference's own fp32 arithmetic has.  Errors below ~5 % of these kinds are inside
    the reference's band by construction -- the kernel-level fp64 tests (test_mixer_backward_kernel_at_awkward_shapes, 1e-4) are the
    gate for those."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), '..', 'tools'))
    from grad_spread import KINDS, grad_spread
    with open(f'{GOLD}/gradnoise.json') as f:
        noise = json.load(f)[name]
    ours = grad_spread(name, manifest, GOLD)
    pow2 = all((v & (v - 1)) == 0 for v in (4 * manifest[name]['h'], 4 * manifest[name].get('w', manifest[name]['h'])))
    rows = {}
    for kd in KINDS:
        o, r = ours[kd], noise[kd]
        rows[kd] = (o['spread_rms'] / r['ref_spread_rms'], o['mean_vs_fp64'] / max(r['ref_vs_fp64'], r['ref_spread_rms']))
    print(name, {k.split('.', 1)[1]: (round(a, 2), round(b, 2)) for k, (a, b) in rows.items()})
    for kd, (rs, rm) in rows.items():
        assert rs <= 2.0, (name, kd, 'spread', rs)
        assert rm <= (2.0 if pow2 else 3.0), (name, kd, 'mean', rm)


@pytest.mark.parametrize('C,h,K,B', [(4, 32, 4, 3), (8, 16, 2, 2)])
def test_split_backward_is_bitwise_the_whole_backward(C, h, K, B):
    """lgteun_backward(LG_FLAG_BWD_LGT) then lgteun_backward(LG_FLAG_BWD_DATA) -- the sequence Engine.train_step uses with
    world > 1 so that the LGT bucket's all-reduce overlaps the K data-step backwards -- leaves exactly the gradient buffer of the
    one-call backward (the LGT's input gradient has to survive in the workspace between the calls)."""
    from gpu_helpers import make_module
    from lgteun_amd._lib import LG_FLAG_BWD_DATA, LG_FLAG_BWD_LGT, LG_FLAG_DROPOUT, LG_FLAG_FAITHFUL, LG_FLAG_SAVE
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(B, C, h, h, seed=9, kind='dn'))
    net = make_module(C, K)
    eng = net.engine()
    flags, seed = LG_FLAG_FAITHFUL | LG_FLAG_SAVE | LG_FLAG_DROPOUT, 1234      # dropout ON: the two calls must redraw the same masks
    out, saved = eng.forward_raw(ms, pan, flags, seed)
    dout = torch.sign(out - gt) / out.numel()
    whole = torch.zeros_like(eng.gflat)
    eng.backward_raw(saved, dout, whole, flags, seed)
    split = torch.zeros_like(eng.gflat)
    eng.backward_raw(saved, dout, split, flags | LG_FLAG_BWD_LGT, seed)
    (a0, b0), (a1, b1) = eng.live_ranges
    assert torch.equal(split[a1:b1], whole[a1:b1])             # bucket 1 (last stage's LGT) is final after the first call ...
    assert float(split[a0:b0].abs().max()) == 0.0               # ... and nothing of the shared bucket has been touched yet
    # what a rank does in between: other work on the stream (here an unrelated forward with another batch size)
    with torch.no_grad():
        net(ms[:1], pan[:1])
    eng.backward_raw(saved, dout, split, flags | LG_FLAG_BWD_DATA, seed)
    assert torch.equal(split, whole)
    assert float(whole[a0:b0].abs().max()) > 0


def test_all_ffn_save_modes_give_the_same_step(monkeypatch):
    """The live stage's e = 16 FFN half-blocks keep, for the backward (LG_FFN_SAVE, read at plan creation): '5' gelu / gelu' of both hidden
    tensors + h2 (five 4e-wide tensors per block, the backward evaluates no GELU); '3' the pre-activations h1 / h2 / h3 (the backward
    re-evaluates GELU with the forward's own function); '2' (default) h2 / h3 only -- h1 is re-computed from x by k_ffn1_bwd_xs, which
    also forms dx, dW1 and dW2 on the bf16 matrix pipe in split arithmetic.  Same forward bit for bit; gradients equal to rounding.
    (All three behind the channel-split forward k_ffn_xs, LG_FFN_FWD=xs: the register chain k_ffn_xr of round 6 knows mode '2' only and sums its
    products in another order -- test_register_chain_ffn_train_step_agrees_with_the_channel_split_kernel compares the two forwards.)"""
    from gpu_helpers import make_module
    from lgteun_amd._lib import LG_FLAG_FAITHFUL, LG_FLAG_SAVE
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(3, 4, 32, 32, seed=5, kind='dn'))
    res = {}
    monkeypatch.setenv('LG_FFN_FWD', 'xs')
    for mode in ('5', '3', '2'):
        monkeypatch.setenv('LG_FFN_SAVE', mode)
        net = make_module(4, 2)
        eng = net.engine()
        out, saved = eng.forward_raw(ms, pan, LG_FLAG_FAITHFUL | LG_FLAG_SAVE, 0)
        ws_sum = float(saved[1].view(torch.uint8)[::4093].double().sum())   # fingerprint of the saved activations (the modes keep different tensors)
        dout = torch.sign(out - gt) / out.numel()
        g = torch.zeros_like(eng.gflat)
        eng.backward_raw(saved, dout, g, LG_FLAG_FAITHFUL | LG_FLAG_SAVE, 0)
        res[mode] = (out.clone(), g.clone(), eng, ws_sum)
    assert res['5'][3] != res['3'][3]          # the switch was honoured: the workspaces hold different things
    assert torch.equal(res['5'][0], res['3'][0]) and torch.equal(res['5'][0], res['2'][0])
    g5, eng = res['5'][1], res['5'][2]
    assert not torch.equal(g5, torch.zeros_like(g5))
    for mode in ('3', '2'):
        gm = res[mode][1]
        for i in eng.live_idx:
            o, n = eng.offsets[i], eng.params[i].numel()
            a, b = g5[o:o + n].double(), gm[o:o + n].double()
            assert float((a - b).norm()) <= 2e-6 * float(a.norm()) + 1e-12, (mode, eng.names[i], float((a - b).norm()), float(a.norm()))
    monkeypatch.delenv('LG_FFN_FWD', raising=False)
    monkeypatch.delenv('LG_FFN_SAVE', raising=False)


def test_register_chain_ffn_train_step_agrees_with_the_channel_split_kernel(monkeypatch):
    """round 6: the default forward of the e = 16 FFN half-blocks is k_ffn_xr (k_ffn_xr.hip); rounds 2 - 5's k_ffn_xs stays as LG_FFN_FWD=xs.  Same
    arithmetic, another summation order: the whole train step (faithful, saving h2 / h3 for the same backward kernels) gives the same
    output to 2e-6 and the same gradient to 2e-5 of its norm (per tensor 3e-2 of the tensor's scale: the cancelling-sum kinds amplify the
    forward's rounding, tests/test_gpu_benchsize.py::test_cancelling_sum_gradient_kinds_band_against_band)"""
    from gpu_helpers import make_module
    from lgteun_amd._lib import LG_FLAG_FAITHFUL, LG_FLAG_SAVE
    ms, pan, gt = (T(a).cuda() for a in dw.make_inputs(3, 4, 32, 32, seed=5, kind='dn'))
    res = {}
    for fwd in ('xr', 'xs'):
        monkeypatch.setenv('LG_FFN_FWD', fwd)
        net = make_module(4, 2)
        eng = net.engine()
        out, saved = eng.forward_raw(ms, pan, LG_FLAG_FAITHFUL | LG_FLAG_SAVE, 0)
        dout = torch.sign(out - gt) / out.numel()
        g = torch.zeros_like(eng.gflat)
        eng.backward_raw(saved, dout, g, LG_FLAG_FAITHFUL | LG_FLAG_SAVE, 0)
        res[fwd] = (out.clone(), g.clone(), eng)
    monkeypatch.delenv('LG_FFN_FWD', raising=False)
    assert not torch.equal(res['xr'][0], res['xs'][0])      # the switch was honoured
    assert rel_l2(res['xr'][0].cpu(), res['xs'][0].cpu()) < 2e-6
    ga, gb, eng = res['xr'][1].double(), res['xs'][1].double(), res['xr'][2]
    assert float((ga - gb).norm()) <= 2e-5 * float(gb.norm())
    for i in eng.live_idx:
        o, n = eng.offsets[i], eng.params[i].numel()
        a, b = ga[o:o + n], gb[o:o + n]
        assert float((a - b).abs().max()) <= 3e-2 * max(float(b.abs().max()), 1e-12), (eng.names[i], float((a - b).abs().max()), float(b.abs().max()))


def test_two_autograd_graphs_keep_their_own_activations():
    """two forwards with the same batch size before either backward (summed / consistency losses): each graph
    owns its saved activations, so the gradients equal those of the two graphs run one after the other"""
    from gpu_helpers import make_module
    net = make_module(4, 2)
    a = [T(x).cuda() for x in dw.make_inputs(2, 4, 8, 8, seed=1, kind='smooth')]
    b = [T(x).cuda() for x in dw.make_inputs(2, 4, 8, 8, seed=2, kind='smooth')]

    def grads_of(loss_fn):
        for p in net.parameters():
            p.grad = None
        loss_fn().backward()
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    l1 = torch.nn.functional.l1_loss
    ga = grads_of(lambda: l1(net(a[0], a[1]), a[2]))
    gb = grads_of(lambda: l1(net(b[0], b[1]), b[2]))

    def both():
        ya = net(a[0], a[1])
        yb = net(b[0], b[1])          # same B: would have overwritten graph a's workspace
        return l1(ya, a[2]) + l1(yb, b[2])
    gab = grads_of(both)
    assert set(gab) == set(ga)
    for k in ga:
        want = ga[k] + gb[k]
        assert float((gab[k] - want).abs().max()) <= 1e-6 * max(float(want.abs().max()), 1e-3), k


# goldens on which the DEFAULT kernel combination draws an angle() branch-cut flip that other, equally accurate roundings do not (recorded, not
# hidden: the kernel-level fp64 tests are the primary gate of each kernel's arithmetic)
BRANCH_CUT_FLIPS_OF_THE_DEFAULT = {}


def test_split_bf16_gemms_are_as_close_to_fp64_as_fp32_arithmetic(manifest, monkeypatch):
    """The FFN GEMMs and (round 5) the four products of the local mixer run on the 16-bit matrix pipe in split arithmetic
    (csrc/split_bf16.h: three bf16 pieces, six piece products; P V: two f16 pieces, three products; fp32 accumulation -- per dot product
    at least as accurate as an fp32 fma chain, tools/micro/split_bf16_gemm.hip, tools/micro/attn_m_check.hip).  Condition for that being
    the HEADLINE arithmetic: on every whole-net reference golden the output is no further from the reference's fp64
    result than 2x what fp32 arithmetic gives -- fp32 arithmetic being the reference's own fp32 run (manifest rel_fp32_vs_fp64) or this
    build's exact fp32 kernels (LG_FFN_IMPL=strip: v_mfma_f32_16x16x4_f32, LG_ATTN_FWD=valu: fp32 FMAs), whichever is further.

    The comparison has to be made ACROSS several roundings of each arithmetic: the FFT mixer's angle() branch cut turns a 1e-7 perturbation
    of a near-negative-real bin into a 1e-5 .. 3e-4 output change, so two equally accurate fp32 evaluations of one net land 100 - 1000x
    apart from fp64 on inputs that have such a bin -- which of them flips is chance (profiles/r05_err_vs_fp64.txt: net_c4_k4_p128 is
    1.1e-7 in some kernel combinations and 1.6e-5, the reference's own fp32 value, in others; net_c4_k2_p32 3.0e-4 in one and 1.6e-7 in
    the rest).  So each arithmetic is evaluated with both local-mixer kernels and both FFT-mixer kernels (four different roundings of the
    same function): the split arithmetic's best evaluation has to be within 2x of the fp32 evaluations' range (the worst of this build's
    four exact-fp32 evaluations and the reference's own fp32 run -- all of them samples of the same flip lottery); every single evaluation
    still has to meet the 1e-3 parity gate.  (Until the real-input FFT kernels arrived the criterion was best-against-best over two
    roundings; changing the FFT's rounding moved grad_c8_k4_p128's flips and showed that two samples are too few for that form.)"""
    from gpu_helpers import make_module
    names = [n for n, m in manifest.items() if n.startswith('net_') or (n.startswith('grad_') and 'w' in m)]
    assert len(names) >= 11
    rows = []
    for name in names:
        m, g = manifest[name], load_gold(name)
        ms, pan, _ = (T(a).cuda() for a in dw.make_inputs(m['B'], m['C'], m['h'], m.get('w', m['h']), seed=m['seed'], kind=m['kind']))
        err = {}
        for impl in ('split', 'strip'):
            for attn in ('m', 'valu'):
                for fft in ('real', 'full'):
                    for k, v in (('LG_FFN_IMPL', 'strip' if impl == 'strip' else None), ('LG_ATTN_FWD', 'valu' if attn == 'valu' else None),
                                 ('LG_FFT', 'full' if fft == 'full' else None)):
                        if v is None:
                            monkeypatch.delenv(k, raising=False)
                        else:
                            monkeypatch.setenv(k, v)       # read once per plan: a fresh module builds a fresh plan
                    net = make_module(m['C'], m['K'])
                    with torch.no_grad():
                        err[impl, attn, fft] = rel_l2(net(ms, pan).cpu().numpy(), g['out_fp64'])
        splits = sorted(v for k, v in err.items() if k[0] == 'split')  # the product arithmetic (its default kernels: ('split', 'm', 'real'))
        split = splits[0]
        exact = max(v for k, v in err.items() if k[0] == 'strip')
        bar = 2.0 * max(exact, m['rel_fp32_vs_fp64'])
        rows.append((name, err, m['rel_fp32_vs_fp64']))
        assert max(err.values()) < 1e-3, rows[-1]
        assert split <= bar, rows[-1]
        #  the best of four must not be the only one that is held to the bar: the MEDIAN of the four roundings is too (two of four may
        # draw a branch-cut flip, three may not), and the shipped default combination itself unless the golden is listed below with its reason
        assert 0.5 * (splits[1] + splits[2]) <= bar, rows[-1]
        if name not in BRANCH_CUT_FLIPS_OF_THE_DEFAULT:
            assert err['split', 'm', 'real'] <= bar, ('default combination', rows[-1])
        else:
            print(f"{name}: default combination {err['split', 'm', 'real']:.2e} against a bar of {bar:.2e}: {BRANCH_CUT_FLIPS_OF_THE_DEFAULT[name]}")
    monkeypatch.delenv('LG_FFT', raising=False)
    monkeypatch.delenv('LG_FFN_IMPL', raising=False)
    monkeypatch.delenv('LG_ATTN_FWD', raising=False)


@pytest.mark.parametrize('C', [4, 8])
def test_matrix_pipe_mixer_is_as_close_to_fp64_as_the_fp32_kernel(C, monkeypatch):
    """The same criterion at kernel level, where no branch cut interferes: the mixer half-block of both levels through k_attn_m (split
    arithmetic on the matrix pipe) and through k_attn (fp32 FMAs), each against an fp64 evaluation of proj(cat(local_mixer, o2)) + x in
    which o2 is THIS build's global-mixer output (one kernel, same input: identical in both runs): the matrix-pipe kernel is within
    1.6x of the fp32 kernel's distance from fp64 (measured 1.3 - 1.5x: 2.0 - 2.7e-7 against 1.4 - 2.1e-7), and both are below 1e-6."""
    from gpu_helpers import Ops, make_module
    from helpers import det_params
    from oracle import lgteun_oracle as orc
    E = 4 * C
    P64 = {k: v.double() for k, v in det_params(C, 1).items()}
    rng = np.random.default_rng(11)
    for blk, e, n in ((0, E, 32), (2, 2 * E, 16)):
        x = T((rng.standard_normal((2, n, n, e)) * 1.5 + 0.3).astype(np.float32))
        pre = 'prior_module.0.' + ('encoder_layers.0.0.blocks.0.' if blk == 0 else 'bottleneck.blocks.0.') + '0.fn.'
        got = {}
        for attn in ('m', 'valu'):
            if attn == 'valu':
                monkeypatch.setenv('LG_ATTN_FWD', 'valu')
            else:
                monkeypatch.delenv('LG_ATTN_FWD', raising=False)
            ops = Ops(make_module(C, 1), 32, 32)
            got[attn] = ops.block(0, blk, 1, x.cuda()).cpu().double()
            o2 = ops.block(0, blk, 0, x.cuda()).cpu().double()               # [B, e/2, n, n]
        y = orc.layer_norm(x.double(), P64[pre + 'norm.weight'], P64[pre + 'norm.bias'])
        x1 = orc.local_mixer(P64, pre + 'fn.local_mixer.', y[..., :e // 2])
        cat = torch.cat((x1, o2.permute(0, 2, 3, 1)), dim=-1).permute(0, 3, 1, 2)
        want = x.double() + orc.point_conv(cat, P64[pre + 'fn.proj.weight'], P64[pre + 'fn.proj.bias']).permute(0, 2, 3, 1)
        den = float((want - x.double()).norm())
        em, ev = float((got['m'] - want).norm()) / den, float((got['valu'] - want).norm()) / den
        print(f'C={C} block {blk}: matrix pipe {em:.3e}  fp32 FMAs {ev:.3e}')
        assert ev < 1e-6 and em < 1e-6 and em <= 1.6 * ev, (blk, em, ev)


@pytest.mark.parametrize('C', [4, 8])
@pytest.mark.parametrize('case', ['as_initialised', 'w1_tiny_w2_huge', 'w1_huge_w3_tiny', 'ln_affine_huge', 'w2_dw_huge_w3_tiny', 'w1_vanishing', 'ln_affine_vanishing'])
def test_f16_pair_ffn_holds_fp32_accuracy_at_extreme_operand_scales(C, case, monkeypatch):
    """The fused FFN forward multiplies f16 PAIRS (split_bf16.h NP = 2), whose exponent range is 2^-24 .. 2^16: every operand is scaled by a
    power of two derived from a bound on the block's weights (k_ffn_prep.hip).  The FFN half-block of both levels (e = 16 / 32 at C = 4,
    32 / 64 at C = 8: all three kernels) against an fp64 evaluation with the block's weights pushed far out of f16's range in opposite
    directions (so that the OUTPUT stays O(1) and the residual add does not drown the comparison) -- hidden activations of 1e-4 and of
    1e3 .. 1e6, weights of 1e-5 and of 1e3, LayerNorm outputs of 1e3: the error stays at the level of the three-piece bf16 arithmetic
    (LG_FFN_SPLIT=bf16x3, which has fp32's exponent range), within 2 x of it (+ 1e-7), and below 2e-6 in absolute terms."""
    from gpu_helpers import Ops, make_module
    from oracle import lgteun_oracle as orc
    E = 4 * C
    mult = {'as_initialised': {},
            'w1_tiny_w2_huge': {'net.0.weight': 1e-4, 'net.0.bias': 1e-4, 'net.2.point_conv.weight': 1e4},
            'w1_huge_w3_tiny': {'net.0.weight': 1e3, 'net.0.bias': 1e3, 'net.4.weight': 1e-3},
            'ln_affine_huge': {'norm.weight': 1e3, 'norm.bias': 1e3, 'net.0.weight': 1e-3},
            #  bounds of 1e-35: the operand scales are capped (k_ffn_prep.hip pow2_below), nothing overflows, the output is finite
            'w1_vanishing': {'net.0.weight': 1e-35}, 'ln_affine_vanishing': {'norm.weight': 1e-35, 'norm.bias': 1e-35},
            'w2_dw_huge_w3_tiny': {'net.2.point_conv.weight': 3e3, 'net.2.point_conv.bias': 3e3, 'net.2.depth_conv.weight': 30.0, 'net.2.depth_conv.bias': 1e5,
                                   'net.4.weight': 1e-5}}[case]
    rng = np.random.default_rng(17)
    for blk, e, n in ((0, E, 32), (2, 2 * E, 16)):
        pre = 'prior_module.0.' + ('encoder_layers.0.0.blocks.0.' if blk == 0 else 'bottleneck.blocks.0.') + '1.fn.'
        x = T((rng.standard_normal((2, n, n, e)) * 1.5 + 0.3).astype(np.float32))
        got = {}
        for split in ('f16x2', 'bf16x3'):
            monkeypatch.setenv('LG_FFN_SPLIT', split)
            net = make_module(C, 1)
            sd = net.state_dict()
            for k, f in mult.items():
                key = pre + (k if k.startswith('norm') else 'fn.' + k)
                sd[key] = sd[key] * f
            net.load_state_dict(sd)
            got[split] = Ops(net, 32, 32).block(0, blk, 2, x.cuda()).cpu().double()
        P64 = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
        y = orc.layer_norm(x.double(), P64[pre + 'norm.weight'], P64[pre + 'norm.bias'])
        want = x.double() + orc.feed_forward(P64, pre + 'fn.', y)
        den = float((want - x.double()).norm())
        e2, e3 = (float((got[s] - want).norm()) / den for s in ('f16x2', 'bf16x3'))
        print(f'C={C} {case} block {blk} (e={e}): |ffn| / |x| = {den / float(x.double().norm()):.2e}   f16 pairs {e2:.3e}   bf16 x 3 {e3:.3e}')
        assert torch.isfinite(got['f16x2']).all(), (case, blk)
        assert e2 < 2e-6 and e2 <= 2.0 * e3 + 1e-7, (case, blk, e2, e3)
    monkeypatch.delenv('LG_FFN_SPLIT', raising=False)


@pytest.mark.parametrize('case', ['as_initialised', 'w2_huge_w1_tiny', 'w1_huge_w2_tiny'])
@pytest.mark.parametrize('gscale', [1e-12, 1.0, 1e8])
def test_f16_pair_ffn_backward_holds_fp32_accuracy_at_extreme_scales(case, gscale, monkeypatch):
    """The pixelwise half of the FFN backward (k_ffn1_bwd_xs, round 5) multiplies f16 PAIRS: LN(x), gelu(h1) and the weights under the
    forward's static power-of-two scales, the gradient operands dh2 / dh1 under scales taken from the launch-wide max |dh2| that the
    spatial half leaves behind (and the bound on dh1 that follows from it).  The FFN half-block of both xs widths (e = 16, 32) against
    fp64 autograd over the oracle, with upstream gradients of 1e-12, 1 and 1e8 times a unit normal and weights pushed out of f16's range
    in opposite directions: for dx and every parameter gradient of the half-block the error stays at the level of the three-piece bf16
    arithmetic (LG_FFN_BWD_SPLIT=bf16x3, which has fp32's exponent range) -- within 2 x of it + 5e-7 of the tensor's norm
    (profiles/r05_ffn_bwd_err.txt: equal to two digits on every tensor).  The pairs are used for the two W2 products only: with pairs in EVERY
    product (the first form of the kernel) dW1 and the LayerNorm gradients came out at 4e-6 .. 1.5e-5 against the triples' 5e-7 -- a pair holds a
    value to 2^-24 relative, a triple exactly, and those sums over pixels cancel to ~1 % of their terms -- which is what this gate is sized to catch."""
    from gpu_helpers import Ops, make_module
    from test_gpu_backward import _oracle_block
    mult = {'as_initialised': {},
            'w2_huge_w1_tiny': {'net.0.weight': 1e-4, 'net.0.bias': 1e-4, 'net.2.point_conv.weight': 1e4},
            'w1_huge_w2_tiny': {'net.0.weight': 1e3, 'net.0.bias': 1e3, 'net.2.point_conv.weight': 1e-3, 'net.4.weight': 1e-3}}[case]
    rng = np.random.default_rng(23)
    for blk, e, n in ((0, 16, 32), (2, 32, 16)):
        pre = 'prior_module.0.' + ('encoder_layers.0.0.blocks.0.' if blk == 0 else 'bottleneck.blocks.0.') + '1.fn.'
        x = T((rng.standard_normal((2, n, n, e)) * 1.5 + 0.3).astype(np.float32))
        dy = T((rng.standard_normal((2, n, n, e)) * gscale).astype(np.float32))
        got = {}
        for split in ('f16x2', 'bf16x3'):
            monkeypatch.setenv('LG_FFN_BWD_SPLIT', split)
            net = make_module(4, 1)
            sd = net.state_dict()
            for k, f in mult.items():
                key = pre + (k if k.startswith('norm') else 'fn.' + k)
                sd[key] = sd[key] * f
            net.load_state_dict(sd)
            ops = Ops(net, 32, 32)
            dx, grads = ops.block_bwd(0, blk, 2, x.cuda(), dy.cuda())
            names = [nm for nm in ops.eng.names if nm.startswith(pre)]
            got[split] = {'dx': dx.double().cpu(), **{nm: ops.grad_of(grads, nm).double().cpu() for nm in names}}
        P64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
        want_dx, want_g = _oracle_block(P64, 4, blk, 2, x.double(), dy.double())
        want = {'dx': want_dx.detach(), **{k: v for k, v in want_g.items() if k.startswith(pre)}}
        assert len(want) == 11 and set(want) == set(got['f16x2'])
        for k, ref in want.items():
            assert torch.isfinite(got['f16x2'][k]).all(), (case, gscale, blk, k)
            rn = float(ref.norm())
            e2, e3 = (float((got[sp][k].reshape(ref.shape) - ref).norm()) for sp in ('f16x2', 'bf16x3'))
            assert e2 <= 2.0 * e3 + 5e-7 * rn + 1e-30, (case, gscale, blk, k, e2 / max(rn, 1e-300), e3 / max(rn, 1e-300))
    monkeypatch.delenv('LG_FFN_BWD_SPLIT', raising=False)


@pytest.mark.gpu
@pytest.mark.parametrize('which', [1, 2])
def test_uneven_work_split_of_the_bench_shape_changes_no_result(which):
    """round 6: at exactly two resident workgroups per CU (32 pairs of 128 x 128 at e = 16) the first workgroup of a CU gets more of the work --
    10 : 6 strip steps in k_ffn_xr, 5 : 3 window quads in k_attn_m, 9 : 7 tiles in k_ffn1_bwd_xs -- and 16 pairs take the even partition.  Half-block
    forward and backward (which = 1: mixer, 2: feed_forward) of the 32-pair batch against the same samples as two 16-pair calls: y and dx are per-pixel
    results and must be BITWISE the same; parameter gradients are sums over all pixels (per-workgroup partial sums in a different partition): the two
    halves' sum to rounding"""
    from gpu_helpers import Ops, make_module
    net = make_module(4, 1)
    ops = Ops(net, 128, 128)
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((32, 128, 128, 16)).astype(np.float32)).cuda()
    y = ops.block(0, 0, which, x).clone()
    dx, g = ops.block_bwd(0, 0, which, x, dy)
    dx, g = dx.clone(), g.clone()
    ya = ops.block(0, 0, which, x[:16].contiguous()).clone()
    yb = ops.block(0, 0, which, x[16:].contiguous()).clone()
    assert torch.equal(y[:16], ya) and torch.equal(y[16:], yb)
    dxa, ga = ops.block_bwd(0, 0, which, x[:16].contiguous(), dy[:16].contiguous())
    dxa, ga = dxa.clone(), ga.clone()
    dxb, gb = ops.block_bwd(0, 0, which, x[16:].contiguous(), dy[16:].contiguous())
    assert torch.equal(dx[:16], dxa) and torch.equal(dx[16:], dxb)
    gs = (ga.double() + gb.double())
    live = g.abs() > 0
    assert bool(live.any())
    err = float((g.double() - gs).norm() / gs.norm())
    assert err < 2e-6, err
