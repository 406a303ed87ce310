"""A process group of ONE rank on backend "nccl" (= RCCL on ROCm) on the box's single MI355X (tests/test_gpu_ddp_engine.py).

Started by tests/conftest.py at session start as a fresh child process, BEFORE the pytest process touches the GPU.  The first 8-GPU
run must not also be the first ncclCommInitRank of this code: here the REAL product path -- Pansharpening.attach_ddp
+ Engine.train_step -- issues its collectives (the weight broadcast, the stream-ordered all-reduce of the default path, the two
ordered buckets of LG_DDP_OVERLAP=serial) on an RCCL communicator, with `force=True` so that a world of one does not skip them.
With one rank a SUM all-reduce is the identity: gradients and weights must be BITWISE those of an unattached engine from the same seed.
Everything observable goes to <outdir>/rccl1.npz (reference: nn.DataParallel's implicit reduce, models/base/base_model.py:91-100)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

C, K, H_MS, B, STEPS = 4, 2, 16, 2, 3      # PAN 64 x 64


def steps(net, ms, pan, gt, n):
    import lgteun_amd
    opt = lgteun_amd.FusedAdam(net.parameters(), lr=1.5e-3)
    opt.dropout = False
    eng = net.engine()
    grads = []
    for _ in range(n):
        eng.train_step(ms, pan, gt, opt)
        grads.append(eng.gflat.detach().cpu().numpy().copy())
    torch.cuda.synchronize()
    return np.stack(grads), eng.flat.detach().cpu().numpy().copy()


def main():
    outdir = sys.argv[1]
    import torch.distributed as dist
    from gpu_helpers import make_module
    from lgteun_amd import ddp
    from oracle import detweights as dw
    torch.cuda.set_device(0)
    os.environ['LGTEUN_FORCE_PG'] = 'nccl'
    rank, world, _ = ddp.init_from_env()
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == 'nccl'
    ms, pan, gt = (torch.from_numpy(a).cuda() for a in dw.make_inputs(B, C, H_MS, H_MS, seed=78, kind='smooth'))
    res = {}
    # the unattached engine: no collective anywhere
    g_ref, w_ref = steps(make_module(C, K, salt=0), ms, pan, gt, STEPS)
    for name, env in (('default', '0'), ('serial', 'serial')):
        os.environ['LG_DDP_OVERLAP'] = env
        net = make_module(C, K, salt=0)
        eng = net.attach_ddp(force=True)                 # dist.broadcast of the flat weights on the RCCL communicator
        assert eng.force_collectives and eng.buckets is not None and eng.world == 1
        assert eng.buckets[False].serial == (env == 'serial')
        g, w = steps(net, ms, pan, gt, STEPS)
        res[f'{name}_grads_equal'] = np.array(int(np.array_equal(g, g_ref)))
        res[f'{name}_weights_equal'] = np.array(int(np.array_equal(w, w_ref)))
        res[f'{name}_gmax'] = np.array(float(np.abs(g).max()))
    # a bare broadcast / all-reduce pair on the flat buffers, and what the step costs with and without the collective
    t = torch.arange(1024, device='cuda', dtype=torch.float32)
    dist.broadcast(t, 0)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    res['bare_ok'] = np.array(int(torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32))))
    res['librccl_mapped'] = np.array(int(any('librccl' in line for line in open('/proc/self/maps'))))
    import lgteun_amd
    # The pytest session and the two data-parallel ranks use the same GPU while this runs: alternate short bursts of the two engines and keep the
    # fastest burst of each (one long run per engine measured whoever happened to share the card with it)
    os.environ['LG_DDP_OVERLAP'] = '0'
    engines = {}
    for name, attach in (('plain', False), ('rccl', True)):
        net = make_module(C, K, salt=0)
        eng = net.attach_ddp(force=True) if attach else net.engine()
        opt = lgteun_amd.FusedAdam(net.parameters(), lr=1.5e-3)
        for _ in range(5):
            eng.train_step(ms, pan, gt, opt)
        engines[name] = (net, eng, opt)
    torch.cuda.synchronize()
    times = {'plain': float('inf'), 'rccl': float('inf')}
    for _ in range(8):
        for name, (net, eng, opt) in engines.items():
            t0 = time.perf_counter()
            for _ in range(10):
                eng.train_step(ms, pan, gt, opt)
            torch.cuda.synchronize()
            times[name] = min(times[name], (time.perf_counter() - t0) / 10)
    res['ms_plain'], res['ms_rccl'] = np.array(times['plain'] * 1e3), np.array(times['rccl'] * 1e3)
    print(f"one-rank nccl group: librccl mapped {int(res['librccl_mapped'])}, step {times['plain'] * 1e3:.3f} ms plain / {times['rccl'] * 1e3:.3f} ms with the collective", flush=True)
    np.savez(os.path.join(outdir, 'rccl1.npz'), **res)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
