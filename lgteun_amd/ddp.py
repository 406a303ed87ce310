"""Batch-sharded data parallelism for the LGTEUN hot path: one process per GPU, torch.distributed (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" on CPU for tests).  Replaces the reference's single-process nn.DataParallel
(models/base/base_model.py:91-100).  The path shards naturally: samples are independent, the only batch
reduction is the L1 mean, so each rank computes its share of the GLOBAL-mean gradient and the flat gradient buffer is
SUM-all-reduced.

Default: ONE collective per step over the span of the flat gradient buffer that covers every live range, issued in stream
order behind the backward (`dist.all_reduce`, not async: RCCL enqueues it on its own stream and makes the compute stream wait
-- no host block; gloo blocks the host until its worker thread is done).  The same call for every backend, so the path the
2-rank tests cover is the path an 8-GPU RCCL run takes.  The live gradients are 0.4 MB (C=4) ... 1.1 MB (C=8) inside a
1.6 ... 8.6 MB buffer whose dead-stage slots are zero: the collective is latency-bound (tens of microseconds of a 7.5 ms
step), so reducing the zeros along costs less than packing the two live ranges would.

Opt-in (`LG_DDP_OVERLAP=1` / `GradBuckets(..., overlap=True)`): two asynchronous buckets -- the last stage's LGT, reduced
while the K data-step backwards still run, and the shared data-module + eta tensors.  It buys < 1 % of a step and is the
form that stalled a 4-ranks-on-one-GPU gloo rehearsal in round 2, so it stays off until an 8-GPU
RCCL run has shown it."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    rank, world, local_rank = env_world()
    if (world > 1 or os.environ.get('LGTEUN_FORCE_PG')) and not dist.is_initialized():
        if world == 1:      # LGTEUN_FORCE_PG=nccl|gloo: a process group of ONE rank (RCCL on the single MI355X of a test box)
            backend = os.environ['LGTEUN_FORCE_PG']
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_bounds(n_global, rank, world):
    """rank r takes samples [r*B/N, (r+1)*B/N) (equal shards)"""
    if n_global % world:
        raise ValueError(f'global batch {n_global} not divisible by world size {world}')
    per = n_global // world
    return rank * per, (rank + 1) * per


def broadcast_flat(flat, src=0, group=None, force=False):
    """identical initial weights on every rank (force: also in a group of one rank -- the single-GPU RCCL check)"""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.broadcast(flat, src=src, group=group)


def overlap_requested():
    """LG_DDP_OVERLAP = 1: two asynchronous buckets (both outstanding behind the data-step backwards); = serial: the same two-bucket
    ordering (LGT backward -> LGT bucket -> data-step backwards -> shared bucket -> Adam) with the first bucket waited for before the
    second is started -- at most ONE outstanding work, the form that is known to complete on every transport"""
    v = os.environ.get('LG_DDP_OVERLAP', '0')
    return 'serial' if v == 'serial' else v == '1'


class GradBuckets:
    """SUM all-reduce of the live part of a flat gradient buffer.  `ranges` = the live ranges, in the reference's graph
    [(a0, b0) shared + eta, (a1, b1) last stage's LGT]; in 'chained' mode [(0, total)]."""

    def __init__(self, ranges, group=None, overlap=None):
        self.ranges = list(ranges)
        self.group = group
        ov = overlap_requested() if overlap is None else overlap
        self.overlap = bool(ov)
        self.serial = ov == 'serial'      # the LGT bucket is waited for in front of the shared bucket (one outstanding work at a time)
        self.span = (min(a for a, _ in self.ranges), max(b for _, b in self.ranges))
        self._pending = []

    # ---- default: one stream-ordered collective --------------------------------------------------
    def all_reduce(self, flat_grad):
        a, b = self.span
        dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group)

    # ---- opt-in overlap: one asynchronous collective per live range -------------------------------
    def start(self, flat_grad, which):
        a, b = self.ranges[which]
        self._pending.append(dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        pending, self._pending = self._pending, []
        for w in pending:
            w.wait()
