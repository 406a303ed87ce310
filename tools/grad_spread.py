"""GPU probe: the band of THIS build's L1 gradients of the cancelling-sum kinds under the same four seeded +-1-ulp input
nudges the reference's own fp32 gradients were put through (tools/gen_goldens.py round4 -> tests/golden/gradnoise.json), per case and kind:
   spread   relative L2 (over the kind's live tensors, on the fp64 gradient's scale) between a nudged run and the un-nudged one, 4 draws
   mean     distance of the MEAN over the 5 draws from the reference's fp64 gradient
next to the reference's numbers.  tests/test_gpu_benchsize.py gates band against band with the same function (lgteun grad_spread below).
   python tools/grad_spread.py [case ...]      -> text for profiles/r05_grad_spread.txt"""
import json
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, R + '/tests')
import numpy as np

KINDS = ('global_mixer.conv_amp.0.bias', 'global_mixer.conv_pha.0.bias', 'global_mixer.conv_amp.0.weight', 'global_mixer.conv_pha.0.weight',
         'local_mixer.pos_emb')


def nudge(a, rng):      # tools/gen_goldens.py round4, verbatim arithmetic: every value to a neighbouring float, direction drawn per value
    up = rng.integers(0, 2, a.shape).astype(bool)
    return np.where(up, np.nextafter(a, np.float32(4.0)), np.nextafter(a, np.float32(-4.0))).astype(np.float32)


def grad_spread(name, manifest, gold_dir, env=None):
    """{kind: dict(spread=[4], spread_rms, mean_vs_fp64, base_vs_fp64)} of this build's gradients in golden case `name`"""
    import torch
    from gpu_helpers import make_module
    from lgteun_amd import FusedAdam
    from oracle import detweights as dw
    m = manifest[name]
    ms, pan, gt = dw.make_inputs(m['B'], m['C'], m['h'], m.get('w', m['h']), seed=m['seed'], kind=m['kind'])
    g64 = np.load(f'{gold_dir}/grad64_{name[5:]}.npz')

    def grads(ms_, pan_):
        net = make_module(m['C'], m['K'])
        opt = FusedAdam(net.parameters(), lr=0.0)
        opt.dropout = False
        eng = net.engine()
        eng.train_step(torch.from_numpy(ms_).cuda(), torch.from_numpy(pan_).cuda(), torch.from_numpy(gt).cuda(), opt)
        out = {}
        for i in eng.live_idx:
            n, o, p = eng.names[i], eng.offsets[i], eng.params[i]
            if n.endswith(KINDS):
                out[n] = eng.gflat[o:o + p.numel()].view(p.shape).cpu().numpy().astype(np.float64)
        return out
    base = grads(ms, pan)
    rng = np.random.default_rng(1000 + m['seed'])
    runs = [grads(nudge(ms, rng), nudge(pan, rng)) for _ in range(4)]
    res = {}
    for kd in KINDS:
        ks = [k for k in base if k.endswith(kd)]
        t64 = {k: g64['g64/' + k.replace('.', '/')] for k in ks}
        den = sum(float((t64[k] ** 2).sum()) for k in ks) ** 0.5
        sp = [sum(float(((r[k] - base[k]) ** 2).sum()) for k in ks) ** 0.5 / den for r in runs]
        mean = {k: (base[k] + sum(r[k] for r in runs)) / 5.0 for k in ks}
        res[kd] = dict(spread=sp, spread_rms=float(np.sqrt(np.mean(np.square(sp)))),
                       mean_vs_fp64=sum(float(((mean[k] - t64[k]) ** 2).sum()) for k in ks) ** 0.5 / den,
                       base_vs_fp64=sum(float(((base[k] - t64[k]) ** 2).sum()) for k in ks) ** 0.5 / den)
    return res


def main():
    from conftest import GOLD
    man = json.load(open(GOLD + '/manifest.json'))
    noise = json.load(open(GOLD + '/gradnoise.json'))
    cases = sys.argv[1:] or sorted(noise)
    print('case / kind'.ljust(58), 'ours: spread rms'.rjust(17), 'ref: spread rms'.rjust(16), 'ratio'.rjust(6), '|', 'ours: mean-fp64'.rjust(16), 'ours: one run'.rjust(14),
          'ref: one run'.rjust(13), 'mean / max(ref)'.rjust(16))
    for name in cases:
        ours = grad_spread(name, man, GOLD)
        for kd in KINDS:
            o, r = ours[kd], noise[name][kd]
            print(f'{name} {kd.split(".", 1)[1]}'.ljust(58), f"{o['spread_rms']:17.3e}", f"{r['ref_spread_rms']:16.3e}", f"{o['spread_rms'] / r['ref_spread_rms']:6.2f}", '|',
                  f"{o['mean_vs_fp64']:16.3e}", f"{o['base_vs_fp64']:14.3e}", f"{r['ref_vs_fp64']:13.3e}", f"{o['mean_vs_fp64'] / max(r['ref_vs_fp64'], r['ref_spread_rms']):16.2f}", flush=True)


if __name__ == '__main__':
    main()
