"""One rank of the 2-process Engine.train_step data-parallel check (tests/test_gpu_ddp_engine.py).

Started by tests/conftest.py at session start -- as fresh child processes, BEFORE the pytest process touches the GPU -- with
RANK / WORLD_SIZE / MASTER_* in the environment; both ranks share the box's one MI355X and talk over gloo (the collective
calls are the ones RCCL serves on an 8-GPU node: torch.distributed all_reduce / broadcast on device tensors).

Each rank runs the REAL product path: Pansharpening.attach_ddp() + Engine.train_step on its shard of a fixed global batch
(dropout off), 3 Adam steps.  Rank 0 then repeats the run in a single process on the concatenated batch.  Everything
observable is written to <outdir>/rank<r>.npz for the test to compare (SURVEY 8e equivalence test; replaces the reference's
nn.DataParallel reduce, models/base/base_model.py:91-100)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

C, K, H_MS, B_GLOBAL, STEPS = 4, 2, 16, 4, 3      # PAN 64 x 64


def run(net, ms, pan, gt, n_steps):
    import lgteun_amd
    opt = lgteun_amd.FusedAdam(net.parameters(), lr=1.5e-3)
    opt.dropout = False
    eng = net.engine()
    out = {}
    for it in range(n_steps):
        loss = eng.train_step(ms, pan, gt, opt)
        out[f'loss{it}'] = loss.detach().cpu().numpy().copy()
        if it == 0:
            out['gflat0'] = eng.gflat.detach().cpu().numpy().copy()
    out['weights'] = eng.flat.detach().cpu().numpy().copy()
    out['ranges'] = np.array(eng.live_ranges)
    return out


def main():
    outdir = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch.distributed as dist
    from gpu_helpers import make_module
    from lgteun_amd import ddp
    from oracle import detweights as dw

    torch.cuda.set_device(0)
    ddp.init_from_env('gloo')
    ms, pan, gt = (torch.from_numpy(a).cuda() for a in dw.make_inputs(B_GLOBAL, C, H_MS, H_MS, seed=77, kind='smooth'))
    a, b = ddp.shard_bounds(B_GLOBAL, rank, world)
    # every rank starts from DIFFERENT weights (salt = rank): attach_ddp must broadcast rank 0's
    net = make_module(C, K, salt=rank)
    net.attach_ddp()
    res = run(net, ms[a:b].contiguous(), pan[a:b].contiguous(), gt[a:b].contiguous(), STEPS)
    res['world'] = np.array(net.engine().world)
    # the attachment survives a rebuilt engine (.to() re-creates the parameters)
    net.to('cuda:0')
    res['world_after_to'] = np.array(net.engine().world)
    dist.barrier()
    if rank == 0:
        single = make_module(C, K, salt=0)
        single.engine().local_only = True          # a deliberate single-process run inside the initialised group
        ref = run(single, ms, pan, gt, STEPS)
        res.update({'single_' + k: v for k, v in ref.items()})
        # an UNATTACHED engine inside a live process group must refuse to train silently unsynchronised
        lone = make_module(C, K, salt=0)
        try:
            run(lone, ms[:2], pan[:2], gt[:2], 1)
            res['unattached_raised'] = np.array(0)
        except RuntimeError as e:
            res['unattached_raised'] = np.array(int('attach_ddp' in str(e)))
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
