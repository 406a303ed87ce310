"""N > 1 path on CPU: world_size-2 gloo run of the DDP bucket logic (lgteun_amd.ddp + the flat live-range layout).
Each rank computes ITS SHARE of the global-mean L1 gradient with the oracle on its batch shard, scatters it into the flat
gradient buffer at the engine's offsets and SUM-all-reduces it -- the default single stream-ordered collective, and the opt-in
form with two asynchronous buckets outstanding at once; the result must equal the single-process
gradient on the concatenated batch (SURVEY 8e equivalence test), and dead-stage slots must stay untouched."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import det_params, state_shapes
from oracle import detweights as dw
from oracle import lgteun_oracle as orc

C, K, H, B = 4, 2, 8, 4


def _flat_grads(P, ms, pan, gt, n_global, mode='faithful'):
    from lgteun_amd.engine import canonical_names, flat_layout
    names = canonical_names(C, K)
    shapes = state_shapes(C, K)
    numels = [int(np.prod(shapes[n])) if len(shapes[n]) else 1 for n in names]
    offs, total, live_idx, live_ranges = flat_layout(names, numels, K)
    if mode == 'chained':       # every tensor is live: one bucket = the whole flat buffer (Engine._live[True])
        live_idx, live_ranges = list(range(len(names))), [(0, total)]
    out = orc.forward(P, ms, pan, K, mode=mode)
    loss = (out - gt).abs().sum() / n_global          # this rank's share of the global mean
    loss.backward()
    flat = torch.zeros(total)
    for i in live_idx:
        g = P[names[i]].grad
        flat[offs[i]:offs[i] + g.numel()] = g.reshape(-1)
    return flat, live_ranges, float(loss)


def _worker(rank, world, port, q, mode, overlap):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from lgteun_amd import ddp
    r, w, _ = ddp.init_from_env('gloo')
    assert (r, w) == (rank, world)
    ms, pan, gt = dw.make_inputs(B, C, H, H, seed=21, kind='smooth')
    a, b = ddp.shard_bounds(B, rank, world)
    T = torch.from_numpy
    P = det_params(C, K, requires_grad=True)
    n_global = B * C * 4 * H * 4 * H
    flat, ranges, loss = _flat_grads(P, T(ms[a:b]), T(pan[a:b]), T(gt[a:b]), n_global, mode)
    # weights broadcast from rank 0 must leave identical buffers identical
    wflat = torch.cat([v.detach().reshape(-1) for v in P.values()])
    before = wflat.clone()
    ddp.broadcast_flat(wflat, 0)
    assert torch.equal(wflat, before)
    buckets = ddp.GradBuckets(ranges, overlap=overlap)
    assert buckets.overlap == bool(overlap) and buckets.serial == (overlap == 'serial')
    if overlap:
        # opt-in form (LG_DDP_OVERLAP=1): two ASYNCHRONOUS collectives outstanding at once, the LGT bucket first (on the GPU path it
        # overlaps the data-step backwards), then unrelated work, then finish()
        buckets.start(flat, 1)
        if buckets.serial:                 # 'serial': the LGT bucket is in before the shared bucket starts (one outstanding work)
            assert len(buckets._pending) == 1
            buckets.finish()
        buckets.start(flat, 0)
        assert len(buckets._pending) == (1 if buckets.serial else 2)
        _ = torch.ones(1000).sum()
        buckets.finish()
        assert not buckets._pending
    else:
        buckets.all_reduce(flat)     # the default: ONE stream-ordered collective over the span of the live ranges (Engine.train_step)
    lt = torch.tensor([loss], dtype=torch.float64)
    dist.all_reduce(lt)
    if rank == 0:
        q.put((flat.numpy(), float(lt.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode,overlap', [('faithful', False), ('faithful', True), ('faithful', 'serial'), ('chained', False)])
def test_two_rank_gradients_equal_single_process(mode, overlap):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got, loss2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ms, pan, gt = dw.make_inputs(B, C, H, H, seed=21, kind='smooth')
    T = torch.from_numpy
    P = det_params(C, K, requires_grad=True)
    want, ranges, loss1 = _flat_grads(P, T(ms), T(pan), T(gt), B * C * 4 * H * 4 * H, mode)
    assert abs(loss1 - loss2) < 1e-6
    want = want.numpy()
    assert np.abs(got - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0) + 1e-7
    if mode == 'chained':
        from lgteun_amd.engine import canonical_names, flat_layout
        shapes = state_shapes(C, K)
        names = canonical_names(C, K)
        _, _, _, d3 = flat_layout(names, [int(np.prod(shapes[n])) if len(shapes[n]) else 1 for n in names], K)
        assert np.abs(got[d3[0][1]:d3[1][0]]).max() > 0      # the first stage's LGT now trains
        return
    (a0, b0), (a1, b1) = ranges
    assert np.all(got[b0:a1] == 0.0)                 # dead-stage slots never receive a gradient
    assert np.abs(got[a1:b1]).max() > 0 and np.abs(got[a0:b0]).max() > 0


def test_shard_bounds_and_layout():
    from lgteun_amd import ddp
    from lgteun_amd.engine import canonical_names, flat_layout
    assert [ddp.shard_bounds(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        ddp.shard_bounds(10, 0, 4)
    names = canonical_names(4, 4)
    shapes = state_shapes(4, 4)
    offs, total, live_idx, ranges = flat_layout(names, [int(np.prod(shapes[n])) if len(shapes[n]) else 1 for n in names], 4)
    assert all(o % 4 == 0 for o in offs) and len(live_idx) == 12 + 4 + 119
    assert ranges[0][0] == 0 and ranges[1][1] == total and ranges[0][1] <= ranges[1][0]
    assert ddp.GradBuckets(ranges).span == (0, total)          # the default bucket: one span over every live range
    assert ddp.GradBuckets(ranges).overlap is False            # two asynchronous buckets are opt-in (LG_DDP_OVERLAP=1)


# ---- sharded evaluation when world does not divide the set: every image is scored exactly once --------------------------
class _EvalSet(torch.utils.data.Dataset):
    """7 tiny 'images' whose PSNR against the target differs per image (the fused image IS the input here: see _EvalRunner).
    ids: 'unique' img0 .. img6; 'collide' img0, img1, img2, img0, .. (PSDataset's image_id is a file-name prefix: two directories can hold
    the same one -- ADVICE r5); 'none' no image_id key at all"""
    N = 7

    def __init__(self, ids='unique'):
        self.ids = ids

    def __len__(self):
        return self.N

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(100 + i)
        tgt = torch.rand(4, 16, 16, generator=g) * 2047
        item = dict(input_lr=tgt + (i + 1) * 3.0 * torch.randn(4, 16, 16, generator=g), input_pan=torch.zeros(1, 16, 16), target=tgt)
        if self.ids != 'none':
            item['image_id'] = f'img{i % 3}' if self.ids == 'collide' else f'img{i}'
        return item


def _eval_runner(loader, rank, world, work):
    from lgteun_amd.base_model import Base_model
    from lgteun_amd.compat import Config

    class _EvalRunner(Base_model):          # the runner's evaluation loop around a stand-in model (no GPU here): output = its input
        def get_model_output(self, input_batch):
            return input_batch['input_lr']
    cfg = Config(dict(work_dir=work, datas='GF-2', bit_depth=11, norm_input=True, loss_cfg=dict(rec_loss=dict(type='l1', w=1.0))))
    r = _EvalRunner(cfg, None, None, None, loader)
    r.add_module('core_module', torch.nn.Linear(1, 1))
    r.rank, r.world = rank, world
    return r


def _eval_worker(rank, world, port, q, work, padded, ids='unique'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from lgteun_amd import dataset as ds, ddp
    ddp.init_from_env('gloo')
    smp = ds.ShardedSampler(_EvalSet.N, rank, world, shuffle=False, pad=padded)
    loader = torch.utils.data.DataLoader(_EvalSet(ids), batch_size=2, sampler=smp)
    out = _eval_runner(loader, rank, world, work).test(iter_id=0, ref=True)
    if rank == 0:
        q.put((out, len(smp)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('padded,ids', [(False, 'unique'), (True, 'unique'), (True, 'collide'), (False, 'collide'), (True, 'none'), (False, 'none')])
def test_two_rank_evaluation_scores_every_image_once(tmp_path, padded, ids):
    """7 images on 2 ranks: the evaluation sampler (pad=False) gives 4 + 3; a PADDED sampler hands image 0 to both ranks and test()
    drops the second copy -- told by its DATASET INDEX (the sampler's), not by image_id: with equal ids on different images ('collide': two
    directories) or no ids at all ('none') every image still counts exactly once.  Either way mean / std equal the
    one-process evaluation's (the reference: base_model.py:267-352)."""
    single = _eval_runner(torch.utils.data.DataLoader(_EvalSet(ids), batch_size=2), 0, 1, str(tmp_path / 's')).test(iter_id=0, ref=True)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q, str(tmp_path / 'd'), padded, ids)) for r in range(2)]
    for p in procs:
        p.start()
    got, n0 = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert n0 == 4
    assert set(got) == set(single) == {'PSNR', 'SSIM', 'Q', 'SAM', 'ERGAS'}
    for k in single:      # the rows arrive in another order (rank-major), so the sums differ in the last bits only
        assert got[k] == pytest.approx(single[k], rel=1e-12, abs=1e-12), k
