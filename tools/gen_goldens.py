"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference)
on CPU in the build container.  Fixtures hold only inputs-by-seed and expected outputs (plain
arrays); weights are regenerated everywhere from oracle/detweights.py.  Run:

    python tools/gen_goldens.py            # writes tests/golden/
    python tools/gen_goldens.py --check    # additionally checks oracle/lgteun_oracle.py against them

Never runs on the GPU box (there is no /root/reference there).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from _ref_import import import_reference  # noqa: E402
from oracle import detweights as dw  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def build_ref(R, C, K, dtype=torch.float32, salt=0):
    net = R.Pansharpening(R.Config(ms_chans=C), None, stage=K)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = dw.fill_state_dict(shapes, salt=salt, dtype=np.float64)
    net = net.to(dtype)
    net.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in sd.items()})
    net.eval()
    return net, shapes


def t(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def round2(R, manifest):
    """Round-2 fixtures: eval forward + L1 gradients of the reference at the BENCH size (128x128 PAN, K=4; C=4 and C=8),
    at BASELINE configs[4]'s shape (C=8, 256x256 PAN, K=8) and at two PAN sizes that are not powers of two (80x48, 208x176:
    the build container's FFT leaves +0 in the imaginary part of the purely real bins there, which pins the angle()=+pi
    convention the HIP mixer uses at every size).  Each file also carries the reference's own fp32-vs-fp64 gradient error."""
    cases = [
        dict(name='grad_c4_k4_p128', C=4, K=4, B=1, h=32, w=32, kind='smooth', seed=21),
        dict(name='grad_c8_k4_p128', C=8, K=4, B=1, h=32, w=32, kind='smooth', seed=22),
        dict(name='grad_c4_k2_p80x48', C=4, K=2, B=1, h=20, w=12, kind='smooth', seed=23),
        dict(name='grad_c4_k2_p208x176', C=4, K=2, B=1, h=52, w=44, kind='smooth', seed=24),
        dict(name='grad_c8_k8_p256', C=8, K=8, B=1, h=64, w=64, kind='smooth', seed=25),
    ]
    for cs in cases:
        C, K, B, h, w = cs['C'], cs['K'], cs['B'], cs['h'], cs['w']
        ms, pan, gt = dw.make_inputs(B, C, h, w, seed=cs['seed'], kind=cs['kind'])
        res = {}
        for dt in (torch.float32, torch.float64):
            net, _ = build_ref(R, C, K, dtype=dt)
            y = net(t(ms, dt), t(pan, dt))
            loss = torch.nn.L1Loss()(y, t(gt, dt))
            loss.backward()
            grads = {k: p.grad.numpy() for k, p in net.named_parameters() if p.grad is not None}
            none_names = [k for k, p in net.named_parameters() if p.grad is None]
            res[dt] = (y.detach().numpy(), float(loss.item()), grads, none_names)
        y32, loss32, g32, none_names = res[torch.float32]
        y64, loss64, g64, _ = res[torch.float64]
        num = sum(float(((g32[k].astype(np.float64) - g64[k]) ** 2).sum()) for k in g32)
        den = sum(float((g64[k] ** 2).sum()) for k in g32)
        # the reference's OWN fp32 noise per tensor (max |g32 - g64| on the tensor's scale, floored at 2e-5): a few sums that cancel
        # to ~1e-5 (FFT-mixer amplitude biases, pos_emb rows) are only known to a few per cent in the reference's fp32 itself
        self_err = np.array([float(np.abs(g32[k] - g64[k]).max() / max(float(np.abs(g64[k]).max()), 2e-5)) for k in sorted(g32)])
        np.savez_compressed(os.path.join(GOLD, cs['name'] + '.npz'), loss=np.array(loss32), loss_fp64=np.array(loss64),
                            out_fp32=y32, out_fp64=y64.astype(np.float32), self_err=self_err,
                            **{k.replace('.', '/'): v for k, v in g32.items()})
        manifest[cs['name']] = dict(cs, salt=0, none_grad_stages=list(range(K - 1)), n_none=len(none_names),
                                    rel_fp32_vs_fp64=float(np.linalg.norm(y32 - y64) / np.linalg.norm(y64)),
                                    grad_rel_fp32_vs_fp64=(num / den) ** 0.5)
        assert all(n.startswith(tuple(f'prior_module.{i}.' for i in range(K - 1))) for n in none_names)
        print(cs['name'], 'loss', loss32, 'live', len(g32), 'none', len(none_names), 'out fp32-vs-fp64', manifest[cs['name']]['rel_fp32_vs_fp64'],
              'grad fp32-vs-fp64', manifest[cs['name']]['grad_rel_fp32_vs_fp64'], flush=True)


R3_KINDS = ('global_mixer.conv_amp.0.bias', 'global_mixer.conv_pha.0.bias', 'global_mixer.conv_amp.0.weight', 'global_mixer.conv_pha.0.weight',
            'local_mixer.pos_emb')


def round3(R, manifest):
    """Round-3 fixtures: the reference's fp64 AND fp32 gradients of the parameter kinds whose L1 gradients are cancelling sums (the FFT
    mixer's per-channel amplitude / phase scale + bias, pos_emb) for the five round-2 cases -> tests/golden/grad64_<case>.npz.  With them
    the train-step test states these kinds' error as a distance to fp64 next to the reference's own fp32 distance, instead of an
    allowance derived from the fp32 golden alone.  The round-2 files are left byte-identical."""
    for name in ('grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c4_k2_p80x48', 'grad_c4_k2_p208x176', 'grad_c8_k8_p256'):
        cs = manifest[name]
        C, K, B, h, w = cs['C'], cs['K'], cs['B'], cs['h'], cs['w']
        ms, pan, gt = dw.make_inputs(B, C, h, w, seed=cs['seed'], kind=cs['kind'])
        out = {}
        for dt, tag in ((torch.float64, 'g64'), (torch.float32, 'g32')):
            net, _ = build_ref(R, C, K, dtype=dt)
            loss = torch.nn.L1Loss()(net(t(ms, dt), t(pan, dt)), t(gt, dt))
            loss.backward()
            for k, p in net.named_parameters():
                if p.grad is not None and k.endswith(R3_KINDS):
                    out[tag + '/' + k.replace('.', '/')] = p.grad.numpy().astype(np.float64)
        old = np.load(os.path.join(GOLD, name + '.npz'))
        for k in [k for k in out if k.startswith('g32/')]:       # the fp32 run is the one the round-2 file already holds
            assert np.array_equal(out[k].astype(np.float32), old[k[4:]]), k
        np.savez_compressed(os.path.join(GOLD, 'grad64_' + name[5:] + '.npz'), **{k: v for k, v in out.items() if k.startswith('g64/')})
        print('grad64_' + name[5:], len([k for k in out if k.startswith('g64/')]), 'tensors', flush=True)


def round4(R, manifest):
    """Round-4 fixture: how well the REFERENCE'S OWN fp32 arithmetic knows the cancelling-sum gradient kinds -> tests/golden/gradnoise.json.
    For each of the five cases and kinds: (a) `ref_vs_fp64`, the distance of the reference's fp32 gradients to its fp64 gradients (relative
    L2 over the kind's live tensors; the number round 3's gate was built on), and (b) `ref_spread`, how far the reference's fp32 gradients
    MOVE when every input value is nudged to a neighbouring float (4 draws of random +-1 ulp on ms and pan, seeded), on the same scale.
    (b) is 5 ... 40 x (a) in the cases where (a) is small: the network is piecewise continuous (torch.angle's branch cut, abs() of the irfft2
    output) and these kinds cancel to ~1e-5 of their terms, so one fp32 evaluation lands anywhere in a band of width (b) around the fp64
    value and a small (a) is a lucky draw, not a property of the arithmetic.  The gate of tests/test_gpu_benchsize.py is per case
    3 x max(a, largest of b) per kind and 5 x the same per tensor."""
    out = {}
    for name in ('grad_c4_k4_p128', 'grad_c8_k4_p128', 'grad_c4_k2_p80x48', 'grad_c4_k2_p208x176', 'grad_c8_k8_p256'):
        cs = manifest[name]
        C, K, B, h, w = cs['C'], cs['K'], cs['B'], cs['h'], cs['w']
        ms, pan, gt = dw.make_inputs(B, C, h, w, seed=cs['seed'], kind=cs['kind'])
        g64 = np.load(os.path.join(GOLD, 'grad64_' + name[5:] + '.npz'))

        def grads32(ms_, pan_):
            net, _ = build_ref(R, C, K, dtype=torch.float32)
            loss = torch.nn.L1Loss()(net(t(ms_), t(pan_)), t(gt))
            loss.backward()
            return {k: p.grad.numpy().astype(np.float64) for k, p in net.named_parameters() if p.grad is not None and k.endswith(R3_KINDS)}

        def nudge(a, rng):
            up = rng.integers(0, 2, a.shape).astype(bool)
            return np.where(up, np.nextafter(a, np.float32(4.0)), np.nextafter(a, np.float32(-4.0))).astype(np.float32)

        base = grads32(ms, pan)
        rng = np.random.default_rng(1000 + cs['seed'])
        runs = [grads32(nudge(ms, rng), nudge(pan, rng)) for _ in range(4)]
        out[name] = {'tensors': {}}
        for k in sorted(base):   # the same two numbers per TENSOR, on the tensor's own scale
            t64 = g64['g64/' + k.replace('.', '/')]
            den = float((t64 ** 2).sum()) ** 0.5
            out[name]['tensors'][k] = dict(ref_vs_fp64=float(((base[k] - t64) ** 2).sum()) ** 0.5 / den,
                                           ref_spread=[float(((r[k] - base[k]) ** 2).sum()) ** 0.5 / den for r in runs])
        for kd in R3_KINDS:
            ks = [k for k in base if k.endswith(kd)]
            t64 = {k: g64['g64/' + k.replace('.', '/')] for k in ks}
            den = sum(float((t64[k] ** 2).sum()) for k in ks) ** 0.5
            ref = sum(float(((base[k] - t64[k]) ** 2).sum()) for k in ks) ** 0.5 / den
            sp = [sum(float(((r[k] - base[k]) ** 2).sum()) for k in ks) ** 0.5 / den for r in runs]
            out[name][kd] = dict(ref_vs_fp64=ref, ref_spread=sp, ref_spread_rms=float(np.sqrt(np.mean(np.square(sp)))))
            print(name, kd.ljust(34), f'ref32-vs-fp64 {ref:.3e}   spread under +-1 ulp inputs', ' '.join(f'{v:.3e}' for v in sp), flush=True)
    with open(os.path.join(GOLD, 'gradnoise.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


# the augmentation cases of tests/golden/dataset_utils.npz: (probabilities in the reference's dict order, the U[0,1) draws the call
# consumes: one per probability, then two per selected crop).  tests/test_dataset_cpu.py feeds lgteun_amd.dataset the same draws.
DATASET_CASES = [
    ('none', None, []),
    ('nothing_selected', dict(ud_flip=0.5, lr_flip=0.5, r4_crop=0.3, r2_crop=0.3), [0.9, 0.8, 0.7, 0.6]),
    ('ud', dict(ud_flip=0.5, lr_flip=0.5), [0.1, 0.9]),
    ('lr', dict(ud_flip=0.5, lr_flip=0.5), [0.9, 0.1]),
    ('ud_and_lr', dict(ud_flip=0.5, lr_flip=0.5), [0.1, 0.2]),                     # the later one wins (utils.py:216-219)
    ('r4', dict(ud_flip=0.5, lr_flip=0.5, r4_crop=0.3, r2_crop=0.3), [0.9, 0.9, 0.1, 0.9, 0.55, 0.80]),
    ('r2', dict(ud_flip=0.5, lr_flip=0.5, r4_crop=0.3, r2_crop=0.3), [0.9, 0.9, 0.9, 0.1, 0.30, 0.95]),
    ('all', dict(ud_flip=0.5, lr_flip=0.5, r4_crop=0.3, r2_crop=0.3), [0.1, 0.1, 0.1, 0.1, 0.99, 0.01, 0.49, 0.51]),
]


def dataset_batch():
    rng = np.random.default_rng(2023)
    return dict(input_lr=rng.integers(0, 2048, (2, 4, 8, 8)).astype(np.float32), input_pan=rng.integers(0, 2048, (2, 1, 32, 32)).astype(np.float32),
                target=rng.integers(0, 2048, (2, 4, 32, 32)).astype(np.float32))


def round5():
    """the pure-torch part of the reference's input pipeline (dataset/utils.py:155-263): data_normalize, data_denormalize and
    data_augmentation for fixed random() sequences -> tests/golden/dataset_utils.npz"""
    from _ref_import import import_reference_dataset_utils
    du = import_reference_dataset_utils()
    base = dataset_batch()
    out = {}
    for name, probs, draws in DATASET_CASES:
        seq = iter(draws)
        du.random = lambda: next(seq)          # the module-level numpy.random.random the function draws from
        batch = dict({k: torch.from_numpy(v.copy()) for k, v in base.items()}, image_id=['a', 'b'])
        res = du.data_augmentation(batch, None if probs is None else dict(probs))
        assert next(seq, None) is None, name   # every listed draw was consumed
        for k in base:
            out[f'aug_{name}_{k}'] = res[k].numpy()
    for bits in (11, 10):
        batch = dict({k: torch.from_numpy(v.copy()) for k, v in base.items()}, image_id=['a', 'b'])
        nrm = du.data_normalize(batch, bits)
        assert nrm['image_id'] == ['a', 'b']
        for k in base:
            out[f'norm{bits}_{k}'] = nrm[k].numpy()
        out[f'denorm{bits}_target'] = du.data_denormalize(nrm['target'], bits).numpy()
    np.savez_compressed(os.path.join(GOLD, 'dataset_utils.npz'), **out)
    print('dataset_utils.npz:', len(out), 'arrays')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only-r5', action='store_true', help='write only the round-5 fixture (the reference\'s dataset/utils.py functions)')
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--only-r3', action='store_true', help='write only the round-3 fixtures (fp64 gradients of the cancelling-sum kinds)')
    ap.add_argument('--only-r2', action='store_true', help='write only the round-2 fixtures (bench-size / configs[4] / non-pow2 '
                                                            'gradients); the manifest is merged, round-1 files are left alone')
    ap.add_argument('--only-r4', action='store_true', help='write only the round-4 fixture (the reference fp32 gradients\' own noise band)')
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if args.only_r5:
        round5()
        return
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    manifest = {}
    if args.only_r4:
        with open(os.path.join(GOLD, 'manifest.json')) as f:
            manifest = json.load(f)
        round4(R, manifest)
        return
    if args.only_r3:
        with open(os.path.join(GOLD, 'manifest.json')) as f:
            manifest = json.load(f)
        round3(R, manifest)
        return
    if args.only_r2:
        with open(os.path.join(GOLD, 'manifest.json')) as f:
            manifest = json.load(f)
        round2(R, manifest)
        with open(os.path.join(GOLD, 'manifest.json'), 'w') as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return

    # ---------------------------------------------------------------- per-op goldens (C=4 and C=8)
    for C in (4, 8):
        net, shapes = build_ref(R, C, 1)
        E = 4 * C
        rng = np.random.default_rng(100 + C)
        out = {}
        x_ms = rng.uniform(0, 1, (2, C, 8, 8)).astype(np.float32)
        out['resample_in'] = x_ms
        out['resample_x4'] = R.bmu.sampling_(t(x_ms), 4).numpy()
        out['resample_x2'] = R.bmu.sampling_(t(x_ms), 2).numpy()
        x_big = rng.uniform(0, 1, (2, C, 32, 32)).astype(np.float32)
        out['z_in'] = x_big
        out['resample_half'] = R.bmu.sampling_(t(x_big), 0.5).numpy()
        out['resample_x1'] = R.bmu.sampling_(t(x_big), 1).numpy()
        pan = rng.uniform(0, 1, (2, 1, 32, 32)).astype(np.float32)
        out['pan_in'] = pan
        with torch.no_grad():
            out['D'] = net.D(t(x_big)).numpy()
            out['DT'] = net.DT(t(x_ms)).numpy()
            z = t(x_big)
            ms_term = net.DT(net.D(z) - t(x_ms))
            pan_term = net.RT(net.R(z) - t(pan))
            out['data_step'] = (z - net.eta[0] * (ms_term + pan_term)).numpy()
            lg = net.prior_module[0]
            out['patch_embed'] = lg.patch_embed(t(x_big)).numpy()
            blk = lg.encoder_layers[0][0].blocks[0]
            mixer = blk[0].fn.fn           # LGMixer
            ffn = blk[1].fn.fn
            feat = rng.standard_normal((2, 32, 32, E)).astype(np.float32)
            # negative-DC planes on the global half of sample 1 pin the angle()=pi branch
            feat[1, :, :, E // 2:] -= 1.5
            out['feat_in'] = feat
            out['local_mixer'] = mixer.local_mixer(t(feat[..., :E // 2]).contiguous()).numpy()
            out['global_mixer'] = mixer.global_mixer(t(feat[..., E // 2:]).contiguous()).numpy()
            out['lg_mixer'] = mixer(t(feat)).numpy()
            out['feed_forward'] = ffn(t(feat)).numpy()
            out['lgb'] = lg.encoder_layers[0][0](t(feat)).numpy()       # NCHW out
            out['lgt'] = lg(t(x_big)).numpy()
        np.savez_compressed(os.path.join(GOLD, f'ops_c{C}.npz'), **out)
        manifest[f'ops_c{C}'] = dict(C=C, K=1, salt=0, keys=sorted(out))

    # ---------------------------------------------------------------- whole-net eval forward
    cases = [
        dict(name='net_c4_k2_p32', C=4, K=2, B=2, h=8, kind='dn'),
        dict(name='net_c8_k2_p32', C=8, K=2, B=1, h=8, kind='smooth'),
        dict(name='net_c4_k4_p64', C=4, K=4, B=1, h=16, kind='smooth'),
        dict(name='net_c4_k4_p128', C=4, K=4, B=1, h=32, kind='dn'),
        dict(name='net_c8_k4_p128', C=8, K=4, B=1, h=32, kind='dn'),
        dict(name='net_c4_k2_p256', C=4, K=2, B=1, h=64, kind='smooth'),   # 256x256 PAN: split-FFT path (BASELINE config 5 size)
    ]
    for cs in cases:
        C, K, B, h = cs['C'], cs['K'], cs['B'], cs['h']
        ms, pan, gt = dw.make_inputs(B, C, h, h, seed=19971118, kind=cs['kind'])
        net, _ = build_ref(R, C, K)
        with torch.no_grad():
            y32 = net(t(ms), t(pan)).numpy()
        net64, _ = build_ref(R, C, K, dtype=torch.float64)
        with torch.no_grad():
            y64 = net64(t(ms, torch.float64), t(pan, torch.float64)).numpy()
        import models.base.metrics as mtc
        o = np.transpose(y32[0], (1, 2, 0)).astype(np.float64) * 2047.5
        g = np.transpose(gt[0], (1, 2, 0)).astype(np.float64) * 2047.5
        met = np.array([mtc.psnr(o, g), mtc.sam(o, g), mtc.ergas(o, g)])
        np.savez_compressed(os.path.join(GOLD, cs['name'] + '.npz'), out_fp32=y32,
                            out_fp64=y64.astype(np.float32), metrics=met)
        manifest[cs['name']] = dict(cs, seed=19971118, salt=0,
                                    rel_fp32_vs_fp64=float(np.linalg.norm(y32 - y64) / np.linalg.norm(y64)))
        print(cs['name'], 'fp32-vs-fp64 rel', manifest[cs['name']]['rel_fp32_vs_fp64'], 'metrics', met)

    # ---------------------------------------------------------------- gradients (dropout off: eval())
    for cs in [dict(name='grad_c4_k2_p32', C=4, K=2, B=2, h=8), dict(name='grad_c8_k2_p32', C=8, K=2, B=1, h=8)]:
        C, K, B, h = cs['C'], cs['K'], cs['B'], cs['h']
        ms, pan, gt = dw.make_inputs(B, C, h, h, seed=7, kind='smooth')
        net, _ = build_ref(R, C, K)
        y = net(t(ms), t(pan))
        loss = torch.nn.L1Loss()(y, t(gt))
        loss.backward()
        grads, none_names = {}, []
        for k, p in net.named_parameters():
            if p.grad is None:
                none_names.append(k)
            else:
                grads[k.replace('.', '/')] = p.grad.numpy()
        np.savez_compressed(os.path.join(GOLD, cs['name'] + '.npz'), loss=np.array(loss.item()), **grads)
        manifest[cs['name']] = dict(cs, seed=7, kind='smooth', salt=0, none_grad=none_names)
        print(cs['name'], 'loss', loss.item(), 'live', len(grads), 'none', len(none_names))

    # ---------------------------------------------------------------- runner-level: 3 train_iter
    # (UnlgFormer.train_iter, Adam + StepLR per iteration, modules kept in eval() -> dropout off)
    import logging
    cfg = R.Config(ms_chans=4, work_dir='/tmp/lgteun_gold', datas='GF-2', cuda=False, max_iter=3,
                   loss_cfg={'rec_loss': dict(type='l1', w=1.)},
                   optim_cfg={'core_module': dict(type='Adam', betas=(0.9, 0.999), lr=1.5e-3)},
                   sched_cfg=dict(step_size=2, gamma=0.85),
                   model_cfg={'core_module': dict(stage=2)})
    logger = logging.getLogger('gold')
    runner = R.UnlgFormer(cfg, logger, None, None, None)
    core = runner.module_dict['core_module']
    shapes = {k: tuple(v.shape) for k, v in core.state_dict().items()}
    sd = dw.fill_state_dict(shapes, salt=0)
    core.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    core.eval()
    runner.set_optim()
    runner.set_sched()
    import mmcv
    runner.timer = mmcv.Timer()
    ms, pan, gt = dw.make_inputs(2, 4, 8, 8, seed=11, kind='smooth')
    losses, lrs = [], []
    logged = []
    runner.print_train_log = lambda it, res, freq=10: logged.append(res['full_loss'])
    for it in range(1, 4):
        lrs.append(runner.optim_dict['core_module'].param_groups[0]['lr'])
        runner.train_iter(it, dict(input_lr=t(ms), input_pan=t(pan), target=t(gt), image_id=['a', 'b']))
        runner.sched_dict['core_module'].step()
    final = {k.replace('.', '/'): v.detach().numpy() for k, v in core.state_dict().items()
             if not k.startswith('prior_module.0.')}
    np.savez_compressed(os.path.join(GOLD, 'train3_c4_k2_p32.npz'), losses=np.array(logged), lrs=np.array(lrs),
                        **final)
    manifest['train3_c4_k2_p32'] = dict(C=4, K=2, B=2, h=8, seed=11, kind='smooth', salt=0, step_size=2,
                                        gamma=0.85, lr=1.5e-3)
    print('train3 losses', logged, 'lrs', lrs)

    round2(R, manifest)
    with open(os.path.join(GOLD, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    round3(R, manifest)
    round4(R, manifest)
    round5()
    print('wrote', GOLD)


if __name__ == '__main__':
    main()
