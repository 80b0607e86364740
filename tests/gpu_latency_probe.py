"""dev probe: latency of the reference-default calls (one rotation: 2 trunk streams) through the Trainer API."""
import sys, os, time, contextlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import orc
import synthetic
from trainer import Trainer
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer('reinforcement', 0.5, False, None, False)
sd = synthetic.make_state_dict(orc.state_layout(1), 0)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
tr.model_target.load_state_dict(tr.model.state_dict())
depth, masks = synthetic.heightmap_scene(0)
md = depth * masks[0]
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for R in (1, 16):
    tr.model.gnum_rotations = tr.model.snum_rotations = R
    f_sweep = lambda: tr.forward(depth, md, 0, True, False, -1)
    f_one = lambda: tr.forward(depth, md, 0, True, False, 0)
    print("R=%d: forward sweep %.2f ms, single-rotation forward %.2f ms" % (R, timeit(f_sweep), timeit(f_one)))
with contextlib.redirect_stdout(sys.stderr):
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    bp = lambda: tr.train_batch(depth, md, 0, [3], np.asarray([0.7]))
    t_bp = timeit(bp, 20)
print("one-sample training step (train_batch, 2 trunk streams: fwd + bwd + Adam) %.2f ms" % t_bp)
