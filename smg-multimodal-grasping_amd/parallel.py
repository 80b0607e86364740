"""Data parallelism for the affordance path: one process per GPU, samples (scene x
mask x rotation) sharded across ranks, ONE gradient all-reduce (RCCL over xGMI when
the backend is "nccl"; gloo in the CPU tests) between backward and Adam.

The reference has no distributed code at all; this is the
MI355X-native addition of SURVEY.md 8e.  Samples are independent - BN statistics are
per sample - so there is no other exchange step on the path."""
import os

import torch
import torch.distributed as dist

import smg_hip


def shard(items, rank=None, world=None):
    """Contiguous block partition of a list of work units; the first (len % world)
    ranks get one extra."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    n = len(items)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return items[start:start + base + (1 if rank < extra else 0)]


def grad_segments(head_out, trunk_id, head_id):
    """(offset, count) ranges of the flat gradient buffer one (trunk, head) backward writes:
    6 953 856 + 160 896 floats = 28.5 MB for the reinforcement net."""
    return [smg_hip.trunk_range(head_out, trunk_id), smg_hip.head_range(head_out, head_id)]


def allreduce_flat(flat, segments, average=False):
    """Sum (or average) the given ranges of a flat tensor over all ranks, in place.
    One collective per contiguous range: no bucketing is needed - the ranges ARE the
    buckets (28 MB and 0.6 MB)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size() == 1 and not os.environ.get("SMG_FORCE_ALLREDUCE"):
        return                      # (the env switch lets a 1-GPU box exercise the RCCL call path)
    for off, n in segments:
        view = flat[off:off + n]
        dist.all_reduce(view, op=dist.ReduceOp.SUM)
        if average:
            view.div_(dist.get_world_size())


def allreduce_grads(model, trunk_id, head_id, average=False):
    """grad_sync hook for Trainer.train_batch."""
    allreduce_flat(model.flat_grads(), grad_segments(model.HEAD_OUT, trunk_id, head_id), average)
