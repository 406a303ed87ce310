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
    # the two-bucket ordering north_star names (LGT backward -> LGT bucket -> K data-step backwards -> shared bucket -> Adam; reference
    # base_model.py:91-100 reduces implicitly) on DEVICE tensors, in the form with one outstanding work at a time:
    # same weights, same shard, same step -> the flat gradient must be bitwise the default path's
    net2 = make_module(C, K, salt=0)
    net2.attach_ddp()
    e2 = net2.engine()
    res['overlap_default_gflat'] = run(net2, ms[a:b].contiguous(), pan[a:b].contiguous(), gt[a:b].contiguous(), 1)['gflat0']
    net3 = make_module(C, K, salt=0)
    net3.attach_ddp()
    e3 = net3.engine()
    for bk in e3.buckets.values():
        bk.overlap, bk.serial = True, True
    res['overlap_serial_gflat'] = run(net3, ms[a:b].contiguous(), pan[a:b].contiguous(), gt[a:b].contiguous(), 1)['gflat0']
    res['overlap_serial_weights_equal'] = np.array(int(torch.equal(e2.flat, e3.flat)))
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
    res.update(runner_check(outdir, rank, world, ms, pan, gt))
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), **res)
    dist.barrier()
    dist.destroy_process_group()


class _ListLogger:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))

    error = warning = info


def runner_check(outdir, rank, world, ms, pan, gt):
    """the reference-style RUNNER on two ranks (Base_model.train / save / test; reference models/base/base_model.py:164-204,354-369
    under one process per GPU instead of nn.DataParallel): ONE rank writes the checkpoint and the fused images, every rank waits for
    the files, the logged loss is the GLOBAL mean, and the written checkpoint loads back with weights_only=True"""
    import lgteun_amd
    from lgteun_amd import ddp
    from lgteun_amd.compat import Config
    a, b = ddp.shard_bounds(B_GLOBAL, rank, world)
    work = os.path.join(outdir, 'runner')
    batch = dict(input_lr=ms[a:b] * 2047.5, input_pan=pan[a:b] * 2047.5, target=gt[a:b] * 2047.5, image_id=[f'r{rank}_{i}' for i in range(b - a)])
    cfg = Config(dict(ms_chans=C, work_dir=work, datas='GF-2', cuda=True, max_iter=2, bit_depth=11, norm_input=True, eval_sharded=True,
                      save_freq=1, eval_freq=-1, test_freq=-1, loss_cfg={'rec_loss': dict(type='l1', w=1.)},
                      optim_cfg={'core_module': dict(type='Adam', betas=(0.9, 0.999), lr=1.5e-3)},
                      sched_cfg=dict(step_size=2, gamma=0.85), model_cfg={'core_module': dict(stage=K)}))
    log = _ListLogger()
    torch.manual_seed(100 + rank)                       # different initial weights per rank: set_cuda() must broadcast rank 0's
    runner = lgteun_amd.build_model('UnlgFormer', cfg, log, [batch], [batch], [batch])
    runner.set_cuda()
    runner.set_optim()
    runner.set_sched()
    runner.optim_dict['core_module'].dropout = False
    out = {'runner_rank': np.array(runner.rank), 'runner_world': np.array(runner.world)}
    runner.train_iter(iter_id=10, input_batch=lgteun_amd.base_model.data_normalize(batch, 11), log_freq=10)   # logs on rank 0
    eng = runner.module_dict['core_module'].engine()
    out['runner_local_loss'] = np.array(float(eng._loss.item()))
    out['runner_global_loss'] = np.array(eng.global_loss())
    logged = [ln for ln in log.lines if 'full loss' in ln]       # 'iteration N of M | lr .. | full loss X | time left ..'
    out['runner_logged_loss'] = np.array(float(logged[-1].split('full loss')[1].split('|')[0]) if logged else -1.0)
    path = runner.save(iter_id=1)                       # rank 0 writes, everyone returns behind the barrier
    out['runner_ckpt_exists'] = np.array(int(os.path.exists(path)))
    out['runner_tmp_left'] = np.array(int(os.path.exists(path + '.tmp')))
    ck = torch.load(path, map_location='cpu', weights_only=True)
    out['runner_ckpt_iter'] = np.array(int(ck['iter_num']))
    w_now = torch.cat([v.detach().reshape(-1).cpu() for v in runner.module_dict['core_module'].state_dict().values()])
    w_ck = torch.cat([v.reshape(-1) for v in ck['core_module'].values()])
    out['runner_ckpt_equal'] = np.array(int(torch.equal(w_now, w_ck)))
    runner.test(iter_id=1, save=True, ref=True)         # every rank evaluates ITS share (cfg.eval_sharded: the loaders are per-rank shards) and writes its TIFFs
    out['runner_eval_psnr'] = np.array(runner.eval_results['PSNR_mean'][-1])
    out['runner_eval_n'] = np.array(len(runner.eval_results['PSNR_mean']))
    d = os.path.join(work, 'GF-2', 'test_out1', 'iter_1')
    out['runner_tifs'] = np.array(sorted(os.listdir(d)) if os.path.isdir(d) else [])
    r2 = lgteun_amd.build_model('UnlgFormer', cfg, _ListLogger(), [batch], [batch], [batch])
    r2.load_checkpoint(path)                            # plain-tensor checkpoint: loads without executing anything from the file
    out['runner_reload_iter'] = np.array(int(r2.last_iter))
    return out


if __name__ == '__main__':
    main()
