#!/usr/bin/env python3
"""dev helper: GPU time of the three head passes of the config-3 step, one by one (bf16 storage)."""
import os, sys, contextlib, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "smg-multimodal-grasping_amd"))
import synthetic
from trainer import Trainer
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer('reinforcement', 0.5, False, None, False)
sd = synthetic.make_state_dict(bench.layout_names(), 0)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
R = 16
tr.model.gnum_rotations = tr.model.snum_rotations = R
depth, masks = synthetic.heightmap_scene(0)
on = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
depth_d, mdepth_d, md2_d = on(depth), on(depth * masks[0]), on(depth * (masks[1] + masks[2]))
labels_d = on(synthetic.uniform(0, "bench/labels", R, 0.0, 1.5), np.float32)
lab1_d = on(synthetic.uniform(5, "bench/labels_c3", 1, 0.0, 1.5), np.float32)
tr.model.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
rots = list(range(R))
calls = [lambda: tr.train_batch(depth_d, mdepth_d, 0, rots, labels_d), lambda: tr.train_batch(depth_d, mdepth_d, 1, rots, labels_d),
         lambda: tr.train_batch(depth_d, md2_d, 2, [0], lab1_d)]
for _ in range(2):
    for c in calls: c()
torch.cuda.synchronize()
for k, c in enumerate(calls):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): c()
    e1.record(); torch.cuda.synchronize()
    print("style %d: %.2f ms per pass" % (k, e0.elapsed_time(e1) / 5))
