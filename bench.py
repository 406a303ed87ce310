#!/usr/bin/env python3
"""bench.py -- train image-pairs/sec of the LGTEUN unfolding hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + L1 + backward + Adam (+ StepLR tick) over one synthetic batch already resident in HBM.
Workload (default --config c2) = BASELINE.json configs[1]: GF-2-shaped 4-band 32x32 MS / 128x128 PAN, K=4 stages, 32 pairs per
GPU, executed in FAITHFUL mode (all K LGTs run forward like the reference; backward over the live graph).  Weak scaling: per-GPU
batch fixed.  --config c3 / c5 run BASELINE configs[2] / [4] (8 bands; parity-test cases that a driver run can also record).
Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- the dominant kernel timed live with HIP events on its launch stream (lg_prof_*), against its roof; `traffic`
                  is read from the committed PMC summary of the same command (profiles/r02_bench_bs32_pmc_hbm.csv)
  eval_forward -- eval-mode forward pairs/s on the same batch (faithful and live), SURVEY 8d
  cpu_baseline -- the oracle's CPU train step at the same batch size and its B=1 eval forward (kind "port") on this host's cores,
                  bounded sample (rank 0, N=1 only), with the CPU model string
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md): HBM3E 8 TB/s spec; fp32 matrix (v_mfma_f32_16x16x4_f32) 157.3 TF spec
PEAK_HBM_GBS = 8000.0
PEAK_F32_MFMA_TFLOPS = 157.3

# fp32-equivalent split arithmetic (csrc/split_bf16.h): three f16-pair MFMAs per 16x16x32 block in the forward FFN (round 5; six bf16-piece
# ones in the backward and under LG_FFN_SPLIT=bf16x3) -> what the 16-bit matrix pipe could deliver
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / (6.0 if os.environ.get('LG_FFN_SPLIT') == 'bf16x3' else 3.0)

CONFIGS = {   # BASELINE.json configs[...] that fit one GPU: (C, K, PAN size, pairs per GPU, label)
    'c2': (4, 4, 128, 32, 'BASELINE configs[1]: C=4, MS 32x32, PAN 128x128, K=4, 32 pairs/GPU'),
    'c3': (8, 4, 128, 32, 'BASELINE configs[2]: C=8, MS 32x32, PAN 128x128, K=4, 32 pairs/GPU'),
    'c5': (8, 8, 256, 16, 'BASELINE configs[4]: C=8, MS 64x64, PAN 256x256, K=8, 16 pairs/GPU'),
}
C, K, H, B_PER_GPU = CONFIGS['c2'][:4]
E, P0 = 4 * C, H * H

PROF_EVERY = 4   # live HIP-event timing of the roofline kernel: every 4th step of the timed region
# kernels that make up the "ffn" launch slot (lg_kernel_id LG_K_FFN2): the fused feed_forward half-block, all variants
FFN_KERNELS = ('k_ffn_xr', 'k_ffn_xs', 'k_ffn_x32', 'k_ffn_strip', 'k_ffn_fused', 'k_ffn1_x64', 'k_ffn2_x64')


def _newest_summary(name):
    """profiles/rNN_bench_<name>_pmc_hbm.csv of the latest round that committed one"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_bench_{name}_pmc_hbm.csv')))
    return found[-1] if found else None


PMC_SUMMARIES = {c: _newest_summary(n) for c, n in (('c2', 'bs32'), ('c3', 'c3'), ('c5', 'c5'))}


def traffic_from_profile(kernel, config='c2'):
    """HBM bytes per launch of the roofline kernel from the committed PMC summary (FETCH_SIZE x2 corrected + WRITE_SIZE, separate
    --pmc passes of `python bench.py`; tools/summarize_profiles.py), launch-weighted over the kernel's variants.  None if absent."""
    import csv
    names = {'ffn': FFN_KERNELS, 'attn': ('k_attn<', 'k_attn_m<'), 'fft': ('k_fftmix<',), 'attn_bwd': ('k_attn_bwd_core', 'k_attn_bwd_f'), 'fft_bwd': ('k_fftmix_bwd',)}.get(kernel)
    path = PMC_SUMMARIES.get(config)
    if not names or not path or not os.path.exists(path):
        return None
    num = den = 0.0
    pair = 0          # e = 64: the half-block is two launches (k_ffn1_x64 + k_ffn2_x64) timed as ONE slot -> bytes add, launches do not
    for r in csv.DictReader(open(path)):
        if any(n in r['Kernel_Name'] for n in names):
            num += float(r['HBM_bytes_per_launch']) * int(r['launches'])
            if 'k_ffn1_x64' in r['Kernel_Name']:
                pair += int(r['launches'])
            den += int(r['launches'])
    den -= pair
    return int(num / den) if den > 0 else None


# per-kernel roofline entries: lg_prof id -> (label, kernel-name patterns of the committed PMC / stats summaries)
KERNEL_GROUPS = [
    ('ffn', 'ffn', FFN_KERNELS),
    ('attn', 'attn', ('k_attn<', 'k_attn_m<')),
    ('fft', 'fft', ('k_fftmix<', 'k_fftmix_r<', 'k_fft_rows_fwd', 'k_fft_cols', 'k_fft_rows_inv')),
    ('attn_bwd', 'attn_bwd', ('k_attn_bwd_f', 'k_attn_bwd_core', 'k_attn_bwd_epi')),
    ('ffn_bwd_spatial', 'ffn2_bwd', ('k_ffn_dw_bwd',)),
    ('ffn_bwd_pixel', 'ffn1_bwd', ('k_ffn1_bwd',)),
    ('fft_bwd', 'fft_bwd', ('k_fftmix_bwd',)),
]


def traffic_by_patterns(patterns, config='c2'):
    """HBM bytes per launch (launch-weighted over the matching kernels) from the newest committed PMC summary; None if absent"""
    import csv
    path = PMC_SUMMARIES.get(config)
    if not path or not os.path.exists(path):
        return None
    num = den = 0.0
    for r in csv.DictReader(open(path)):
        if any(n in r['Kernel_Name'] for n in patterns):
            num += float(r['HBM_bytes_per_launch']) * int(r['launches'])
            den += int(r['launches'])
    return int(num / den) if den > 0 else None


def synth_batch(B, rank, device, c=None, h=None):
    """integer DN in [0,2047] / 2047.5 (reference dataset/utils.py:232-249, bit_depth 11); seed configs/unlg_former.py:66"""
    c, h = c or C, h or H
    g = torch.Generator().manual_seed(19971118 + rank)

    def dn(*shape):
        return (torch.randint(0, 2048, shape, generator=g).float() / 2047.5).to(device)
    return dn(B, c, h // 4, h // 4), dn(B, 1, h, h), dn(B, c, h, h)


# what the SQ counters of a committed profile show the roofline kernel to be held by -- per (config, kernel), with the file that says so.
# No entry = no counters were taken for that kernel in that config: the label is then None rather than a guess.
LIMITERS = {('c2', 'ffn'): ('valu-issue (75 % vector-busy at two waves per SIMD)', 'profiles/r06_sq_counters_ffn.txt'),
            ('c2', 'attn_bwd'): ('valu-issue', 'profiles/r05_sq_counters_step.txt'),
            ('c3', 'ffn'): ('valu-issue (49 % vector-active, 17 % matrix-busy, in alternating phases)', 'profiles/r05_sq_counters_c3.txt'),
            ('c5', 'ffn'): ('valu-issue (49 % vector-active, 17 % matrix-busy, in alternating phases)', 'profiles/r05_sq_counters_c5.txt')}


def side_config(name, device, n_steps):
    """side measurement (NOT `value`): the train step of another BASELINE config on a fresh module -- pairs/s, ms per step and the
    average launch of its fused FFN forward (HIP events on every launch of the timed steps), so that the driver's record carries the
    8-band configs next to the headline"""
    import lgteun_amd
    from lgteun_amd import _lib
    from lgteun_amd.compat import Config
    c, k, h, b, label = CONFIGS[name]
    torch.manual_seed(19971118)
    net = lgteun_amd.Pansharpening(Config(ms_chans=c), None, stage=k).to(device)
    net.train()
    eng = net.engine()
    opt = lgteun_amd.FusedAdam(net.parameters(), lr=1.5e-3, betas=(0.9, 0.999))
    ms, pan, gt = synth_batch(b, 0, device, c, h)
    L = _lib.lib()
    for _ in range(3):
        eng.train_step(ms, pan, gt, opt)
    _lib.check(L.lg_prof_enable(_lib.KERNEL_IDS['ffn'], 64 * (n_steps + 1)), 'lg_prof_enable')
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n_steps):
        eng.train_step(ms, pan, gt, opt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n_steps
    tot_ms, n_l = ctypes.c_double(), ctypes.c_int64()
    _lib.check(L.lg_prof_read(ctypes.byref(tot_ms), ctypes.byref(n_l)), 'lg_prof_read')
    L.lg_prof_disable()
    del net, eng, opt
    torch.cuda.empty_cache()
    return dict(value=round(b / dt, 2), unit='image-pairs/sec', ms_per_step=round(dt * 1e3, 3), steps=n_steps, workload=label, mode='faithful',
                dropout=True, ffn_avg_launch_us=round(tot_ms.value / max(n_l.value, 1) * 1e3, 2), ffn_launches=int(n_l.value),
                note='fresh module, HIP events on every fused-FFN launch of the timed steps (event pairs cost ~1 % of a step)')


def algorithmic_per_launch(kernel, B):
    """ALGORITHMIC work of one average launch of `kernel` (SURVEY 8d per-unit figures):
    bytes = what the ideally fused unit moves (reads its input once, writes its output once, fp32); flops = 2 x MAC of
    its convs.  Per LGT a block kernel runs on 4 level-0 blocks (E ch, P0 px) and 1 level-1 block (2E ch, P0/4 px);
    figures are averaged over those 5 launches."""
    px = [(E, P0)] * 4 + [(2 * E, P0 // 4)]
    if kernel == 'ffn':
        # k_ffn_strip (level 0, e=16) / k_ffn_fused (level 1, e=32) = the whole feed_forward half-block: x in, y out (+ planar LN half for the next mixer: e/2)
        byts = sum((2 * e + e // 2) * p * 4 for e, p in px) / 5 * B
        flops = sum((2 * (e * 4 * e + 4 * e * 4 * e + 4 * e * e) + 18 * 4 * e) * p for e, p in px) / 5 * B
        return byts, flops
    if kernel in ('fft', 'attn', 'fft_bwd', 'attn_bwd'):
        # mixer half-block unit: read x (e), write y (e); the two kernels split it by channel half
        byts = sum(2 * e * p * 4 for e, p in px) / 5 * B / 2
        flops = sum((2 * (3 * (e // 2) ** 2 + e * e) + 2 * 2 * 64 * (e // 2)) * p for e, p in px) / 5 * B if kernel == 'attn' else 0.0
        return byts, flops
    if kernel in ('ffn_bwd_spatial', 'ffn_bwd_pixel'):
        # the feed_forward half-block's backward as ONE unit (two launches): read x, dy, the saved h2 / h3 (4e each), write dx; 2 x the forward's
        # flops (every product once towards the input, once towards its weight).  Split: the spatial half owns W3 and the depthwise conv,
        # the pixelwise half W1 and W2 -- bytes: dy + h2 + h3 in, dh2 out | x + dh2 in, dx out
        if kernel == 'ffn_bwd_spatial':
            byts = sum((e + 3 * 4 * e) * p * 4 for e, p in px) / 5 * B
            flops = sum(2 * (2 * 4 * e * e + 18 * 4 * e) * p for e, p in px) / 5 * B
        else:
            byts = sum((2 * e + 4 * e) * p * 4 for e, p in px) / 5 * B
            flops = sum(2 * 2 * (e * 4 * e + 4 * e * 4 * e) * p for e, p in px) / 5 * B
        return byts, flops
    if kernel == 'attn_bwd_unit':
        # the mixer half-block's backward: read x, dy, write dx (the local-mixer kernel's half); 2 x the forward's flops
        byts = sum(3 * e * p * 4 for e, p in px) / 5 * B / 2
        flops = 2 * sum((2 * (3 * (e // 2) ** 2 + e * e) + 2 * 2 * 64 * (e // 2)) * p for e, p in px) / 5 * B
        return byts, flops
    return 0.0, 0.0   # kernels without a per-unit figure in SURVEY 8d: time only


def host_cores():
    """cores this process may actually use: cgroup CPU quota if set, else the affinity mask; capped at 32"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 32))


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    return 'unknown'


def cpu_baseline(cores, config):
    """oracle (CPU restatement, torch CPU fp32; SURVEY 8d "CPU baseline beside it"): train step (forward faithful + L1 + backward +
    Adam) at the bench batch size, >= 3 timed steps after 1 warm-up, and the B=1 eval forward of configs[0]; bounded to ~10-30 s of
    CPU work (the 8-band configs time a smaller batch and say so)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import det_params
    from oracle import lgteun_oracle as orc
    torch.set_num_threads(cores)
    Bc = B_PER_GPU if config == 'c2' else (8 if config == 'c3' else 1)
    P = det_params(C, K, requires_grad=True)
    ms, pan, gt = synth_batch(Bc, 0, 'cpu')
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in P.items()}

    def step(it):
        for v in P.values():
            v.grad = None
        loss = orc.l1_loss(orc.forward(P, ms, pan, K, mode='faithful'), gt)
        loss.backward()
        with torch.no_grad():
            for k, v in P.items():
                if v.grad is None:
                    continue
                p, m1, v1 = orc.adam_step(v, v.grad, mom[k][0], mom[k][1], it, 1.5e-3)
                v.copy_(p)
                mom[k] = (m1, v1)
    step(1)                                   # warm-up
    n = 3
    t0 = time.time()
    for it in range(n):
        step(it + 2)
    dt = (time.time() - t0) / n
    # configs[0]: GF-2 4-band, 32x32 MS / 128x128 PAN, K=4, batch 1, eval forward (the reference's own CPU-runnable case)
    P1 = det_params(4, 4)
    g = torch.Generator().manual_seed(19971118)
    ms1 = torch.randint(0, 2048, (1, 4, 32, 32), generator=g).float() / 2047.5
    pan1 = torch.randint(0, 2048, (1, 1, 128, 128), generator=g).float() / 2047.5
    with torch.no_grad():
        orc.forward(P1, ms1, pan1, 4, mode='faithful')
        t1 = time.time()
        for _ in range(3):
            orc.forward(P1, ms1, pan1, 4, mode='faithful')
        dt1 = (time.time() - t1) / 3
    return dict(value=round(Bc / dt, 3), unit='train image-pairs/sec', cores=cores, kind='port', cpu_model=cpu_model(),
                sample=f'{n} timed train steps (fwd faithful + L1 + bwd + Adam), batch {Bc}, fp32 torch-CPU oracle, after 1 warm-up',
                eval_forward_b1=dict(value=round(1.0 / dt1, 3), unit='eval image-pairs/sec',
                                     sample='3 timed eval forwards (faithful), configs[0]: C=4, 128x128 PAN, K=4, batch 1'))


def main():
    global C, K, H, B_PER_GPU, E, P0
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)   # 0.3 s timed region at configs[1] (20 steps were 0.14 s)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS), help='c2 = BASELINE configs[1] (the metric; default); c3 / c5 = the 8-band configs')
    ap.add_argument('--prof-kernel', default='ffn', help='kernel timed live for the roofline object (default: the dominant one)')
    ap.add_argument('--mode', default='faithful', choices=['faithful', 'live', 'chained'],
                    help="'faithful' = the reference's graph (headline); 'live' / 'chained' are labelled non-headline variants")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-live', action='store_true', help='skip the side measurements (profiling runs)')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'],
                    help="fp32: parity mode (default).  bf16: saved/hidden FFN activations of the backward stored as bf16")
    args = ap.parse_args()
    C, K, H, B_PER_GPU, label = CONFIGS[args.config]
    E, P0 = 4 * C, H * H

    from lgteun_amd import ddp
    rank, world, local_rank = ddp.env_world()
    # watchdog: a stalled rank dumps every thread's Python stack and exits non-zero instead of sitting in a collective until the
    # launcher's time limit kills it without a trace.  Armed by default for multi-rank runs (600 s covers RCCL's first-collective
    # set-up plus the whole run many times over); LG_BENCH_WATCHDOG=N overrides, 0 disables.
    wd = int(os.environ.get('LG_BENCH_WATCHDOG', '600' if world > 1 else '0'))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run', file=sys.stderr)
            sys.exit(2)
    if local_rank >= torch.cuda.device_count():   # rehearsal of N ranks on fewer GPUs (gloo): share devices
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    forced_pg = os.environ.get('LGTEUN_FORCE_PG') if world == 1 else None   # nccl: a ONE-rank RCCL group on a single-GPU box (what the collective costs)
    # librccl prints a version banner on STDOUT when its first communicator is created (measured: five lines in front of the JSON line of
    # the one-rank run); stdout carries exactly ONE JSON line by contract, so file descriptor 1 points at stderr until the communicator
    # exists (a first collective below forces its creation)
    import torch.distributed as dist

    import lgteun_amd
    from lgteun_amd import _lib
    from lgteun_amd.compat import Config

    saved_stdout = None
    try:
        if world > 1 or forced_pg:
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            ddp.init_from_env(os.environ.get('LGTEUN_DDP_BACKEND', 'nccl'))   # nccl = RCCL over xGMI; gloo only for rehearsal
        torch.manual_seed(19971118)
        net = lgteun_amd.Pansharpening(Config(ms_chans=C), None, stage=K).to(device)
        net.mode = args.mode
        net.precision = args.precision
        net.train()
        eng = net.attach_ddp(force=bool(forced_pg)) if (world > 1 or forced_pg) else net.engine()
        if saved_stdout is not None:  # attach_ddp has broadcast the weights: the communicator exists and has said what it had to say
            torch.cuda.synchronize()
    finally:                          # whatever the group set-up raised: file descriptor 1 is stdout again
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    opt = lgteun_amd.FusedAdam(net.parameters(), lr=1.5e-3, betas=(0.9, 0.999))          # configs/unlg_former.py:82-84
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=25900, gamma=0.85)              # :86, stepped every iteration
    ms, pan, gt = synth_batch(B_PER_GPU, rank, device)

    trace = os.environ.get('LG_BENCH_TRACE')   # diagnostic: per-step wall time of every rank (synchronises each step)
    nstep = [0]

    def step():
        t_ = time.perf_counter()
        eng.train_step(ms, pan, gt, opt)
        sched.step()
        if trace:
            torch.cuda.synchronize()
            nstep[0] += 1
            print(f'[rank {rank}] step {nstep[0]} {1e3 * (time.perf_counter() - t_):.1f} ms', file=sys.stderr, flush=True)

    def timed(fn, n_warm, n):
        for _ in range(n_warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    import warnings
    warnings.filterwarnings('ignore', message='Detected call of')
    L = _lib.lib()
    kid = _lib.KERNEL_IDS[args.prof_kernel]
    if kid:   # --prof-kernel none: no HIP events inside the timed region (A/B of their cost; the roofline object is then empty)
        _lib.check(L.lg_prof_enable(kid, 64 * (args.steps + 1)), 'lg_prof_enable')
    for _ in range(args.warmup):
        step()
    if kid:
        L.lg_prof_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # the roofline kernel is timed live with HIP events on every PROF_EVERY-th step of the timed region: an event pair costs ~2 us of
    # stream time, 40 of them per step were 1.1 % of the step (same-box A/B: 7.89 vs 7.80 ms); the sampled launches are still inside
    # the region and `roofline.launches` says how many there were
    t0 = time.perf_counter()
    for it in range(args.steps):
        if kid:
            L.lg_prof_pause(0 if it % PROF_EVERY == 0 else 1)
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if kid:
        L.lg_prof_pause(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tot_ms, n_l = ctypes.c_double(), ctypes.c_int64()
    if kid:
        _lib.check(L.lg_prof_read(ctypes.byref(tot_ms), ctypes.byref(n_l)), 'lg_prof_read')
        L.lg_prof_disable()
    loss = float(eng._loss.item()) * world if world == 1 else None
    side = world == 1 and not args.no_live
    n_side = max(5, args.steps // 2)
    # side measurement (NOT `value`): the same train step with the K-1 dead LGT forwards skipped -- bit-identical outputs,
    # gradients and weights (tests/test_gpu_fullsize.py), i.e. what a user of this framework can run instead
    live = None
    if args.mode == 'faithful' and side:
        net.mode = 'live'
        dt = timed(step, 2, n_side)
        live = dict(value=round(B_PER_GPU / dt, 2), ms_per_step=round(dt * 1e3, 3), note='dead-stage LGT forwards skipped; identical results')
        net.mode = args.mode
    # side measurement (NOT `value`): the faithful step with its dead-stage LGT forwards enqueued on a second stream behind the LGT
    # backward (engine.overlap_dead; bitwise the same step, tests/test_gpu_fullsize.py).  Off by default: the co-running launches
    # stretch the per-kernel durations the roofline object reports.
    overlap = None
    if args.mode == 'faithful' and side:
        net.engine().overlap_dead = True
        dt = timed(step, 2, n_side)
        overlap = dict(value=round(B_PER_GPU / dt, 2), ms_per_step=round(dt * 1e3, 3),
                       note='dead-stage LGT forwards on a second stream beside the data-step backwards + Adam; identical results')
        net.engine().overlap_dead = False
    # second side measurement (NOT `value`): BASELINE configs[1] names bf16 training.  The opt-in throughput mode (plain bf16 MFMA in
    # the FFN forward, bf16 storage of the tensors saved for the backward; everything else fp32) is gated against the default mode in
    # tests/test_gpu_backward.py (>= 50 dB PSNR, gradients within 2e-2); the headline stays the fp32-accurate mode the 1e-3 gate is
    # stated for.
    bf16 = None
    if args.precision == 'fp32' and side:
        net.precision = 'bf16'
        dt = timed(step, 3, n_side)
        bf16 = dict(value=round(B_PER_GPU / dt, 2), ms_per_step=round(dt * 1e3, 3), mode=args.mode,
                    note='precision="bf16" throughput mode: plain bf16 MFMA in the FFN forward + bf16 saved activations; PSNR vs default mode >= 50 dB (tested)')
        net.precision = 'fp32'
    # third side measurement: eval-mode forward (no dropout, nothing saved), the reference's get_model_output path
    evalf = None
    if side:
        net.eval()
        evalf = {}
        with torch.no_grad():
            for mode in ('faithful', 'live'):
                net.mode = mode
                net.faithful_eval = True           # time exactly the requested graph (the module would otherwise skip dead stages in eval)
                dt = timed(lambda: net(ms, pan), 3, n_side)
                evalf[mode] = dict(value=round(B_PER_GPU / dt, 2), ms_per_batch=round(dt * 1e3, 3))
        net.mode = args.mode
        net.train()

    # fourth side measurement: BASELINE configs[2] and configs[4] at their single-GPU shapes (8 bands), fresh modules
    others = None
    if side and args.config == 'c2' and args.mode == 'faithful' and args.precision == 'fp32':
        others = {name: side_config(name, device, 10) for name in ('c3', 'c5')}

    # per-kernel roofline entries: three extra steps per kernel id OUTSIDE the timed region, HIP events on every launch
    # of that id (lg_prof times one id at a time)
    def prof_kernel(name, n_steps=3, fn=None):
        _lib.check(L.lg_prof_enable(_lib.KERNEL_IDS[name], 64 * (n_steps + 1)), 'lg_prof_enable')
        for _ in range(n_steps):
            (fn or step)()
        torch.cuda.synchronize()
        tm, nl = ctypes.c_double(), ctypes.c_int64()
        _lib.check(L.lg_prof_read(ctypes.byref(tm), ctypes.byref(nl)), 'lg_prof_read')
        L.lg_prof_disable()
        return tm.value / max(nl.value, 1) * 1e3, nl.value / n_steps

    by_kernel = None
    if side and args.mode == 'faithful':
        by_kernel = []
        times = {klabel: prof_kernel(kid_name) for klabel, kid_name, _ in KERNEL_GROUPS}
        for klabel, kid_name, pats in KERNEL_GROUPS:
            avg_us_k, per_step = times[klabel]
            byts_k, flops_k = algorithmic_per_launch('attn_bwd_unit' if klabel == 'attn_bwd' else klabel, B_PER_GPU)
            by_kernel.append(dict(kernel=klabel, avg_us=round(avg_us_k, 2), launches_per_step=round(per_step, 1), alg_bytes=int(byts_k), alg_flops=int(flops_k),
                                  hbm_frac=round(byts_k / (avg_us_k * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) if avg_us_k > 0 else None,
                                  mfma_frac=round(flops_k / (avg_us_k * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if avg_us_k > 0 else None,
                                  traffic=traffic_by_patterns(pats, args.config)))
    # the bf16 mode's own roofline: its fused FFN forward against the DENSE bf16 matrix peak it executes on
    if bf16 is not None:
        net.precision = 'bf16'
        for _ in range(2):
            step()
        avg_us_b, _ = prof_kernel('ffn')
        net.precision = 'fp32'
        _, flops_b = algorithmic_per_launch('ffn', B_PER_GPU)
        ach = flops_b / (avg_us_b * 1e-6) / 1e12 if avg_us_b > 0 else 0.0
        bf16['roofline'] = dict(bound='mfma', achieved=round(ach, 2), peak=PEAK_BF16_MFMA_TFLOPS, unit='TFLOP/s', frac=round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                                kernel='fused FFN forward, plain bf16 MFMA (one piece per operand)', avg_launch_us=round(avg_us_b, 2),
                                note='limiter: vector issue (GELU, depthwise conv, LayerNorm), not the bf16 matrix pipe')
    # GPU single-image inference latency next to the paper's Table 1 (25.4 ms / image at K = 4, 13.7 ms at K = 2 on an RTX 3090): configs[0]'s
    # shape, batch 1, every call synchronised
    b1 = None
    if side:
        b1 = {}
        msb, panb, _ = synth_batch(1, 0, device, 4, 128)
        for kk in (4, 2):
            nb1 = lgteun_amd.Pansharpening(Config(ms_chans=4), None, stage=kk).to(device).eval()
            nb1.faithful_eval = True
            for mode in ('faithful', 'live'):
                nb1.mode = mode
                with torch.no_grad():
                    for _ in range(5):
                        nb1(msb, panb)
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    for _ in range(30):
                        nb1(msb, panb)
                        torch.cuda.synchronize()
                    b1[f'K{kk}_{mode}'] = round((time.perf_counter() - t_) / 30 * 1e3, 3)
            del nb1
        b1 = dict(unit='ms per image (batch 1, synchronised per call)', workload='configs[0]: C=4, MS 32x32, PAN 128x128', **b1,
                  paper_rtx3090_ms=dict(K4=25.4, K2=13.7), note='paper Table 1 numbers are the reference on other hardware: context, not a baseline for `vs_baseline`')

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B_PER_GPU * world * args.steps / elapsed
        byts, flops = algorithmic_per_launch(args.prof_kernel, B_PER_GPU)
        avg_us = tot_ms.value / max(n_l.value, 1) * 1e3
        ach_gbs = byts / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
        ach_tf = flops / (avg_us * 1e-6) / 1e12 if avg_us > 0 else 0.0
        f_hbm, f_mfma = ach_gbs / PEAK_HBM_GBS, ach_tf / PEAK_F32_MFMA_TFLOPS
        # the binding roof is the one the kernel sits closer to: fp32 GEMM-bearing units are matrix-core bound
        if f_mfma >= f_hbm:
            roof = dict(bound='mfma', achieved=round(ach_tf, 2), peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=round(f_mfma, 4))
        else:
            roof = dict(bound='hbm', achieved=round(ach_gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(f_hbm, 4))
        pmc_path = PMC_SUMMARIES.get(args.config)
        # `bound` names the ROOF the fraction is taken against (contract: hbm | mfma); `limiter` says what the counters show the kernel
        # is actually held by (profiles/r0N_sq_counters_*): the vector ALU's instruction issue, not either roof
        lim = LIMITERS.get((args.config, args.prof_kernel))
        roof.update(limiter=lim[0] if lim and os.path.exists(os.path.join(ROOT, lim[1])) else None, limiter_source=lim[1] if lim else None,
                    traffic=traffic_from_profile(args.prof_kernel, args.config),
                    traffic_source=os.path.relpath(pmc_path, ROOT) if pmc_path and os.path.exists(pmc_path) else None,
                    kernel=L.lg_kernel_name(kid).decode(), launches=int(n_l.value), timed_every_n_steps=PROF_EVERY,
                    avg_launch_us=round(avg_us, 2), algorithmic_bytes_per_launch=int(byts), algorithmic_flops_per_launch=int(flops),
                    hbm_frac=round(f_hbm, 4), mfma_frac_fp32=round(f_mfma, 4), peak_hbm_GBs=PEAK_HBM_GBS,
                    peak_fp32_mfma_TFLOPs=PEAK_F32_MFMA_TFLOPS,
                    note='peak = the f32 matrix rate (the reference arithmetic\'s dtype). The GEMMs execute as 3 f16 MFMAs per product '
                         f'(fp32-equivalent two-piece split, power-of-two operand scales): their own pipe would allow {PEAK_SPLIT_TFLOPS:.0f} TFLOP/s; the kernel is bound by its '
                         'VALU work (two erf-GELUs per hidden element, depthwise 3x3, LayerNorm, operand splitting), not by either matrix rate')
        metric = 'train image-pairs/sec, GF-2 4-band 128x128, K=4, bs=32/GPU' if args.config == 'c2' else f'train image-pairs/sec, {label}'
        out = dict(metric=metric, value=round(value, 2), unit='image-pairs/sec',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3), higher_is_better=True,
                   scaling='weak', vs_baseline=None,
                   dtype='f32 (GEMMs and the local mixer: split 16-bit MFMA -- f16 pairs in the forward FFN, the W2 products of its backward and P V, bf16 triples elsewhere -- fp32 accumulate; fp32-equivalent)' if args.precision == 'fp32'
                         else 'bf16 / f16 MFMA (FFN, local mixer) + bf16 saved activations, f32 elsewhere', data='synthetic',
                   config=dict(workload=label + ', train step = fwd + L1 + bwd + Adam + StepLR tick', mode=args.mode,
                               global_batch=B_PER_GPU * world, parallelism=f'dp{world}', dropout=True),
                   roofline=roof)
        out['watchdog_s'] = wd
        if world > 1 or forced_pg:
            bk = eng.buckets[False] if eng.buckets else None
            out['ddp'] = dict(backend=dist.get_backend(), bucket_form=('two buckets, LG_DDP_OVERLAP=' + os.environ.get('LG_DDP_OVERLAP', '')) if (bk is not None and bk.overlap)
                              else 'one stream-ordered all-reduce behind the backward')
        if forced_pg:
            out['forced_process_group'] = dict(backend=dist.get_backend(), world=1, note='LGTEUN_FORCE_PG: the per-step gradient all-reduce (and the weight '
                                               'broadcast) run on a one-rank communicator: what the collective call costs on this box, not a scaling point')
        if loss is not None:
            out['final_loss'] = round(loss, 6)
        if live is not None:
            out['live_mode'] = live
        if overlap is not None:
            out['overlap_dead_mode'] = overlap
        if bf16 is not None:
            out['bf16_mode'] = bf16
        if evalf is not None:
            out['eval_forward'] = dict(unit='eval image-pairs/sec', batch=B_PER_GPU, **evalf)
            if b1 is not None:
                out['eval_forward']['b1'] = b1
        if by_kernel is not None:
            out['roofline_by_kernel'] = by_kernel
        if others is not None:
            out['c3_mode'] = others['c3']
            out['c5_mode'] = others['c5']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(host_cores(), args.config)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
