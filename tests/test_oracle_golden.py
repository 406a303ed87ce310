"""Pin oracle/lgteun_oracle.py against outputs of the reference itself (tests/golden/, generated
by tools/gen_goldens.py in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_gold
from helpers import det_params, rel_l2, state_shapes
from oracle import detweights as dw
from oracle import lgteun_oracle as orc

T = torch.from_numpy


@pytest.mark.parametrize('C', [4, 8])
def test_state_dict_surface(C, manifest):
    # 119 tensors per stage + 12 shared + K eta
    for K in (1, 2, 4):
        s = state_shapes(C, K)
        assert len(s) == 119 * K + 12 + K
    tot = sum(int(np.prod(v)) if len(v) else 1 for v in state_shapes(C, 4).values())
    assert tot == {4: 404193, 8: 1079741}[C]          # paper Table 1 / SURVEY §6


@pytest.mark.parametrize('C', [4, 8])
def test_ops_vs_reference(C):
    g = load_gold(f'ops_c{C}')
    P = det_params(C, 1)
    E = 4 * C
    tol = 2e-6
    x_ms, z, pan, feat = T(g['resample_in']), T(g['z_in']), T(g['pan_in']), T(g['feat_in'])
    assert rel_l2(orc.resample(x_ms, 4), g['resample_x4']) < tol
    assert rel_l2(orc.resample(x_ms, 2), g['resample_x2']) < tol
    assert rel_l2(orc.resample(z, 0.5), g['resample_half']) < tol
    assert rel_l2(orc.resample(z, 1), g['resample_x1']) == 0.0
    assert rel_l2(orc.op_D(P, z), g['D']) < tol
    assert rel_l2(orc.op_DT(P, x_ms), g['DT']) < tol
    assert rel_l2(orc.data_step(P, z, x_ms, pan, P['eta.0']), g['data_step']) < tol
    pre = 'prior_module.0.'
    assert rel_l2(orc.patch_embed(P, pre + 'patch_embed.', z), g['patch_embed']) < tol
    bp = pre + 'encoder_layers.0.0.blocks.0.'
    # reference local_mixer returns [(b nW), 64, c]; window merge is LGT.py:207-208
    lm = g['local_mixer'].reshape(2, 4, 4, 8, 8, E // 2).transpose(0, 1, 3, 2, 4, 5).reshape(2, 32, 32, E // 2)
    assert rel_l2(orc.local_mixer(P, bp + '0.fn.fn.local_mixer.', feat[..., :E // 2]), lm) < tol
    # the FFT branch is sensitive to the angle() branch cut: looser
    assert rel_l2(orc.global_mixer(P, bp + '0.fn.fn.global_mixer.', feat[..., E // 2:]), g['global_mixer']) < 1e-4
    assert rel_l2(orc.lg_mixer(P, bp + '0.fn.fn.', feat), g['lg_mixer']) < 1e-4
    assert rel_l2(orc.feed_forward(P, bp + '1.fn.fn.', feat), g['feed_forward']) < tol
    assert rel_l2(orc.lgb(P, pre + 'encoder_layers.0.0.', feat, 2).permute(0, 3, 1, 2), g['lgb']) < 1e-4
    assert rel_l2(orc.lgt(P, pre, z), g['lgt']) < 1e-4


@pytest.mark.parametrize('name', ['net_c4_k2_p32', 'net_c8_k2_p32', 'net_c4_k4_p64', 'net_c4_k4_p128',
                                  'net_c8_k4_p128', 'net_c4_k2_p256'])
def test_whole_net_vs_reference(name, manifest):
    m = manifest[name]
    g = load_gold(name)
    ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m['h'], seed=m['seed'], kind=m['kind'])
    P = det_params(m['C'], m['K'])
    with torch.no_grad():
        y_live = orc.forward(P, T(ms), T(pan), m['K'], mode='live')
    # north_star tolerance: 1e-3 relative fp32; the oracle itself is held much tighter
    assert rel_l2(y_live, g['out_fp32']) < 1e-4
    assert rel_l2(y_live, g['out_fp64']) < 1e-4
    if m['h'] <= 16:
        with torch.no_grad():
            y_f = orc.forward(P, T(ms), T(pan), m['K'], mode='faithful')
        assert torch.equal(y_f, y_live)           # dead stages do not influence the output (D3)
        P64 = det_params(m['C'], m['K'], dtype=torch.float64)
        with torch.no_grad():
            y64 = orc.forward(P64, T(ms).double(), T(pan).double(), m['K'])
        assert rel_l2(y64, g['out_fp64']) < 1e-5
    # PSNR / SAM equal to 3 d.p. (north_star)
    o = np.transpose(y_live[0].numpy(), (1, 2, 0)).astype(np.float64) * 2047.5
    t = np.transpose(gt[0], (1, 2, 0)).astype(np.float64) * 2047.5
    met = np.array([orc.psnr(o, t), orc.sam(o, t), orc.ergas(o, t)])
    assert np.allclose(np.round(met, 3), np.round(g['metrics'], 3), atol=1.1e-3)


@pytest.mark.parametrize('name', ['grad_c4_k2_p32', 'grad_c8_k2_p32'])
def test_gradients_vs_reference(name, manifest):
    m = manifest[name]
    g = load_gold(name)
    ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m['h'], seed=m['seed'], kind=m['kind'])
    P = det_params(m['C'], m['K'], requires_grad=True)
    loss = orc.l1_loss(orc.forward(P, T(ms), T(pan), m['K']), T(gt))
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    none = sorted(k for k, v in P.items() if v.grad is None)
    assert none == sorted(m['none_grad'])
    assert sorted(orc.live_param_names(P, m['K'])) == sorted(k for k in P if k not in none)
    worst = 0.0
    for k, v in P.items():
        if v.grad is None:
            continue
        ref = g[k.replace('.', '/')]
        scale = max(np.abs(ref).max(), 1e-6)
        worst = max(worst, float(np.abs(v.grad.numpy() - ref).max() / scale))
    assert worst < 2e-3, worst


def test_train3_vs_reference(manifest):
    """3 x (forward, L1, backward, Adam, StepLR) -- UnlgFormer.train_iter + Base_model.train cadence."""
    m = manifest['train3_c4_k2_p32']
    g = load_gold('train3_c4_k2_p32')
    ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m['h'], seed=m['seed'], kind=m['kind'])
    P = det_params(m['C'], m['K'], requires_grad=True)
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in P.items()}
    losses, lrs = [], []
    for it in range(3):
        lr = orc.steplr(m['lr'], it, m['step_size'], m['gamma'])
        lrs.append(lr)
        for v in P.values():
            v.grad = None
        loss = orc.l1_loss(orc.forward(P, T(ms), T(pan), m['K']), T(gt))
        loss.backward()
        losses.append(loss.item())
        with torch.no_grad():
            for k, v in P.items():
                if v.grad is None:
                    continue                       # Adam skips params without grad (dead stages)
                p, m1, v1 = orc.adam_step(v, v.grad, mom[k][0], mom[k][1], it + 1, lr)
                v.copy_(p)
                mom[k] = (m1, v1)
    assert np.allclose(lrs, g['lrs'], rtol=1e-12)
    assert np.allclose(losses, g['losses'], rtol=2e-4), (losses, g['losses'])
    for k, v in P.items():
        if k.startswith('prior_module.0.'):
            continue
        assert rel_l2(v.detach(), g[k.replace('.', '/')]) < 5e-3, k


def test_chained_mode_is_the_same_functions_composed_the_intended_way():
    """mode='chained'  is not the reference, so no golden exists: it is pinned through the two functions it
    composes (data_step, lgt -- both pinned above) by writing the composition out by hand."""
    C, K, h = 4, 3, 8
    ms, pan, _ = (T(a) for a in dw.make_inputs(2, C, h, h, seed=9, kind='smooth'))
    P = det_params(C, K)
    z = orc.resample(ms, 4)
    for i in range(K):
        z = orc.lgt(P, f'prior_module.{i}.', orc.data_step(P, z, ms, pan, P[f'eta.{i}']))
    got = orc.forward(P, ms, pan, K, mode='chained')
    assert torch.equal(got, z)
    assert rel_l2(got, orc.forward(P, ms, pan, K, mode='faithful')) > 1e-3
    assert torch.equal(orc.forward(P, ms, pan, 1, mode='chained'), orc.forward(P, ms, pan, 1, mode='faithful'))
    with pytest.raises(ValueError):
        orc.forward(P, ms, pan, K, mode='intended')


def test_reference_noise_band_fixture_covers_every_case_kind_and_tensor(manifest):
    """tests/golden/gradnoise.json (tools/gen_goldens.py --only-r4, the reference itself): for each of the five bench-size / odd-size cases the
    reference's own fp32 distance to its fp64 gradients and its spread under +-1-ulp input nudges, per kind and per live tensor of the
    cancelling-sum kinds -- the numbers the per-case gate of tests/test_gpu_benchsize.py is built on"""
    import json
    import os
    from conftest import GOLD
    noise = json.load(open(os.path.join(GOLD, 'gradnoise.json')))
    kinds = ('global_mixer.conv_amp.0.bias', 'global_mixer.conv_pha.0.bias', 'global_mixer.conv_amp.0.weight', 'global_mixer.conv_pha.0.weight',
             'local_mixer.pos_emb')
    assert sorted(noise) == sorted(['grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c4_k2_p80x48', 'grad_c4_k2_p208x176', 'grad_c8_k8_p256'])
    for name, entry in noise.items():
        g64 = np.load(os.path.join(GOLD, 'grad64_' + name[5:] + '.npz'))
        tensors = {k[4:].replace('/', '.') for k in g64.files}
        assert set(entry['tensors']) == tensors and len(tensors) == 25          # 5 blocks x 5 kinds of the live LGT
        assert all(k.startswith(f"prior_module.{manifest[name]['K'] - 1}.") for k in tensors)
        for kd in kinds:
            e = entry[kd]
            assert 0 < e['ref_vs_fp64'] < 0.2 and len(e['ref_spread']) == 4 and all(0 < v < 0.2 for v in e['ref_spread'])
        for t in entry['tensors'].values():
            assert 0 <= t['ref_vs_fp64'] < 1.0 and len(t['ref_spread']) == 4
    # the finding the gate rests on: where the reference's fp32 sits unusually close to fp64, its own one-ulp spread is an order of magnitude larger
    e = noise['grad_c8_k4_p128']['global_mixer.conv_pha.0.weight']
    assert max(e['ref_spread']) > 20 * e['ref_vs_fp64']
