#!/usr/bin/env python3
"""dev helper: time BASELINE.json config 5's per-GPU share (S = 1824, 4 rotations as one training call, fp32-class mode) - run
under different env settings / SMG_HIP_LIB builds by tools/ab5.sh."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "smg-multimodal-grasping_amd"))
import synthetic
from trainer import Trainer
from oracle import affordance as orc
tr = Trainer('reinforcement', 0.5, False, None, False)
sd = synthetic.make_state_dict(orc.state_layout(1), 0)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
if os.environ.get("AB5_PRE"):        # what bench.py did before: a big 640 batch on the other engine, precision round trips
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    nb = int(os.environ["AB5_PRE"])
    sc = [synthetic.heightmap_scene(200 + k) for k in range(nb)]
    d8 = torch.from_numpy(np.stack([c[0] for c in sc])).to(tr.model._flat_params.device)
    m8 = torch.from_numpy(np.stack([c[0] * c[1][0] for c in sc])).to(tr.model._flat_params.device)
    lab8 = torch.zeros(nb * 16, dtype=torch.float32, device=d8.device)
    tr.train_batch(d8, m8, 0, [list(range(16))] * nb, lab8)
    if os.environ.get("AB5_PREC"):
        tr.model.set_precision('bf16'); tr.train_batch(d8, m8, 0, [list(range(16))] * nb, lab8); tr.model.set_precision('fp32')
if os.environ.get("AB5_REL"):
    import models
    torch.cuda.synchronize(); models.release_engines()
tr.model.gnum_rotations = tr.model.snum_rotations = 32
dev = tr.model._flat_params.device
dbig, mbig = synthetic.heightmap_scene(4, size=640, n_boxes=8)
d = torch.from_numpy(dbig).to(dev); m = torch.from_numpy(dbig * mbig[0]).to(dev)
lab = torch.tensor([0.3, 1.9, 0.1, 0.7], dtype=torch.float32, device=dev)
for _ in range(2): tr.train_batch(d, m, 0, [5, 6, 7, 8], lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): tr.train_batch(d, m, 0, [5, 6, 7, 8], lab)
torch.cuda.synchronize()
print("%.2f ms per config-5 step" % ((time.perf_counter() - t0) / 5 * 1e3))
