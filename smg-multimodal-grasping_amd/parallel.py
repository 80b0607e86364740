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
    via_host = flat.is_cuda and dist.get_backend() == "gloo"      # (tests: two ranks sharing one GPU cannot use RCCL)
    for off, n in segments:
        view = flat[off:off + n]
        if via_host:
            tmp = view.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
            view.copy_(tmp)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM)
        if average:
            view.div_(dist.get_world_size())


def allreduce_grads(model, trunk_id, head_id, average=False):
    """grad_sync hook for Trainer.train_batch."""
    allreduce_flat(model.flat_grads(), grad_segments(model.HEAD_OUT, trunk_id, head_id), average)


class OverlappedGradSync(object):
    """grad_sync hook for Trainer.train_batch that hides the all-reduce under the backward: the engine runs the backward in two
    halves (smg_backward_phase); after the first - head, dense blocks 4 / 3 / 2: 5.9 M of the 7.1 M gradient elements - their
    ranges are all-reduced asynchronously (RCCL on its own stream, ordered behind the first half by the launch stream's event)
    while dense block 1, pool0 and the stem compute; the rest follows and Adam waits for both.  Over gloo with tensors on a
    shared GPU (tests) the collectives go through the host and are synchronous - same results, no overlap."""
    overlapped = True

    def __init__(self, average=False):
        self.average = average
        self.pending = []

    def _ranges(self, model, trunk_id, head_id):
        (t0, tn), head = grad_segments(model.HEAD_OUT, trunk_id, head_id)
        split = smg_hip.trunk_split(model.HEAD_OUT, trunk_id)
        return [(split, t0 + tn - split), head], [(t0, split - t0)]

    def _reduce(self, flat, segments):
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() == 1 and not os.environ.get("SMG_FORCE_ALLREDUCE"):
            return
        if flat.is_cuda and dist.get_backend() == "gloo":
            allreduce_flat(flat, segments, self.average)
            return
        for off, n in segments:
            view = flat[off:off + n]
            self.pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view))

    def start(self, model, trunk_id, head_id):
        self._reduce(model.flat_grads(), self._ranges(model, trunk_id, head_id)[0])

    def finish(self, model, trunk_id, head_id):
        self._reduce(model.flat_grads(), self._ranges(model, trunk_id, head_id)[1])
        for work, view in self.pending:
            work.wait()                      # (makes the launch stream wait for the collective's stream)
            if self.average:
                view.div_(dist.get_world_size())
        self.pending = []


def sweep_sharded(trainer, depth_heightmap, m_depth_heightmap, style=0, is_target=False):
    """The R-rotation Q sweep of code/main.py:165-173 with the rotations sharded over the ranks: rank r
    evaluates a contiguous block of rotations (each rank recomputes the masked stream: one extra trunk pass), the R
    scalars are all-gathered (64 bytes - no other exchange on the forward path) and every rank takes the argmax on the
    host, lowest index on ties like np.argmax (main.py:172).  Returns (q[R] float64, best rotation).
    BN running statistics: every rank applies its own rotations' updates (replicas diverge in the buffers the training-mode
    forward never reads); call it on the target network or re-sync the buffers if they matter."""
    model = trainer.model_target if is_target else trainer.model
    R = model.gnum_rotations if style == 0 else (model.snum_rotations if style == 1 else 1)
    rots = list(range(R))
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    mine = shard(rots) if world > 1 else rots
    q_mine = []
    if mine:
        hm = trainer._heightmaps_to_device(depth_heightmap, m_depth_heightmap)
        num = model.gnum_rotations if style == 0 else (model.snum_rotations if style == 1 else model.gnum_rotations)
        q = model.run(style, [0 if style == 2 else r for r in mine], num, heightmaps=hm, mean=trainer.image_mean, std=trainer.image_std)
        q_mine = [float(v) for v in q.reshape(-1).cpu().numpy().astype("float64")]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, q_mine)
        q_all = [v for p in parts for v in p]
    else:
        q_all = q_mine
    import numpy as np
    q_all = np.asarray(q_all, dtype=np.float64)
    return q_all, int(np.argmax(q_all))
