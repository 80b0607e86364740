"""world_size-2 gloo tests of the data-parallel pieces (run on CPU): the sample
sharding and the flat-gradient all-reduce that bench.py / Trainer.train_batch use on
RCCL.  The collective code is backend-agnostic (torch.distributed.all_reduce on views of
the flat gradient buffer), so gloo exercises the same lines."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO  # noqa: F401  (sets sys.path)

import parallel
import smg_hip


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # sample sharding: 16 rotations of 3 scenes = 48 units over 2 ranks
        units = [(s, r) for s in range(3) for r in range(16)]
        mine = parallel.shard(units)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        # flat gradient all-reduce over the (trunk 1, head 1) segments only
        n = smg_hip.lib().smg_layout_param_floats(1)
        flat = torch.full((n,), float(rank + 1))
        segs = parallel.grad_segments(1, 1, 1)
        parallel.allreduce_flat(flat, segs)
        inside = sum(float(flat[o:o + c].sum()) for o, c in segs)
        total = float(flat.sum())
        # rotation-sharded forward sweep (stub model: Q(r) = a known table), 16 scalars gathered, argmax on every rank
        table = torch.tensor([0.1, -0.3, 0.7, 0.2, 0.7, 0.0, -1.0, 0.5, 0.3, 0.6, -0.2, 0.1, 0.4, 0.69, 0.2, -0.5])

        class _Model(object):
            gnum_rotations = snum_rotations = 16

            def run(self, style, rots, num, heightmaps=None, mean=0.0, std=1.0):
                assert num == 16 and len(rots) == 8
                return table[rots].reshape(-1, 1, 1, 1)

        class _Trainer(object):
            model = model_target = _Model()
            image_mean, image_std = 0.01, 0.03

            def _heightmaps_to_device(self, a, b):
                return None
        q, best = parallel.sweep_sharded(_Trainer(), None, None, style=0)
        # overlapped form: the ranges behind dense block 1 (+ the head) first, the rest after the second half of the backward

        class _Net(object):
            HEAD_OUT = 1

            def __init__(self):
                self.g = torch.full((n,), float(rank + 1))

            def flat_grads(self):
                return self.g
        net, ov = _Net(), parallel.OverlappedGradSync()
        early, late = ov._ranges(net, 1, 1)
        ov.start(net, 1, 1)
        for w, _ in ov.pending:
            w.wait()
        after_start = (sum(float(net.g[o:o + c].sum()) for o, c in early), sum(float(net.g[o:o + c].sum()) for o, c in late))
        ov.finish(net, 1, 1)
        after_finish = sum(float(net.g[o:o + c].sum()) for o, c in early + late)
        if rank == 0:
            out.put((gathered, segs, inside, total, n, q, best, early, late, after_start, after_finish, float(net.g.sum())))
    finally:
        dist.destroy_process_group()


def test_shard_and_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, segs, inside, total, n, q, best, early, late, after_start, after_finish, ov_total = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    units = [(s, r) for s in range(3) for r in range(16)]
    assert gathered[0] + gathered[1] == units and len(gathered[0]) == 24           # contiguous, complete, balanced
    seg_elems = sum(c for _, c in segs)
    assert seg_elems == 6953856 + 160896                                            # 28.5 MB per style
    assert inside == pytest.approx(3.0 * seg_elems)                                 # 1 + 2 summed inside the segments
    assert total == pytest.approx(3.0 * seg_elems + 1.0 * (n - seg_elems))          # untouched elsewhere (rank 0 holds 1.0)
    assert q.shape == (16,) and q.dtype == np.float64 and abs(q[13] - 0.69) < 1e-6    # rank order = rotation order
    assert best == 2                                                                  # tie 0.7 / 0.7: lowest index, like np.argmax
    # OverlappedGradSync: early + late ranges tile the same (trunk, head) segments; start() reduces the early ones only
    n_early, n_late = sum(c for _, c in early), sum(c for _, c in late)
    assert n_early + n_late == seg_elems and n_late < n_early and late[0][0] == segs[0][0]
    assert after_start[0] == pytest.approx(3.0 * n_early) and after_start[1] == pytest.approx(1.0 * n_late)
    assert after_finish == pytest.approx(3.0 * seg_elems) and ov_total == pytest.approx(total)


def test_shard_uneven():
    items = list(range(10))
    parts = [parallel.shard(items, r, 4) for r in range(4)]
    assert sum(parts, []) == items
    assert [len(p) for p in parts] == [3, 3, 2, 2]
    assert parallel.shard([], 0, 2) == []


def test_allreduce_is_noop_without_process_group():
    flat = torch.arange(10.0)
    parallel.allreduce_flat(flat, [(0, 10)])
    assert torch.equal(flat, torch.arange(10.0))
