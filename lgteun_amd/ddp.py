"""Batch-sharded data parallelism for the LGTEUN hot path: one process per GPU, torch.distributed (backend "nccl" is RCCL
over xGMI on ROCm; "gloo" on CPU for tests).  Replaces the reference's single-process nn.DataParallel
(models/base/base_model.py:91-100).  The path shards naturally: samples are independent, the only batch
reduction is the L1 mean, so each rank computes its share of the GLOBAL-mean gradient and the flat live-gradient
ranges are SUM-all-reduced -- two buckets: the last stage's LGT (ready first, reduced while the K data-step backwards
still run) and the shared data-module + eta tensors (a few hundred bytes)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    rank, world, local_rank = env_world()
    if world > 1 and not dist.is_initialized():
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


def broadcast_flat(flat, src=0, group=None):
    """identical initial weights on every rank"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


class GradBuckets:
    """SUM all-reduce of the live ranges of a flat gradient buffer.  `ranges` = [(a0,b0) shared+eta, (a1,b1) last LGT]."""

    def __init__(self, ranges, group=None):
        self.ranges = list(ranges)
        self.group = group
        self._pending = []

    def start(self, flat_grad, which):
        a, b = self.ranges[which]
        # gloo on device tensors (the N-ranks-on-one-GPU rehearsal of bench.py / tests, never the production transport) runs its
        # collectives on host threads that copy through pinned memory: with two of them outstanding beside a stream that keeps
        # receiving work, a 4-rank rehearsal stalled for seconds and then for good (every rank in finish(); blocking collectives: fine).
        # RCCL collectives are stream-ordered and stay asynchronous.  LG_DDP_SYNC=1 forces the blocking form for any backend.
        if os.environ.get('LG_DDP_SYNC') == '1' or (flat_grad.is_cuda and dist.get_backend(self.group) == 'gloo'):
            dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group)
            return
        w = dist.all_reduce(flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(w)

    def finish(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def all_reduce(self, flat_grad):
        for i in range(len(self.ranges)):
            self.start(flat_grad, i)
        self.finish()
