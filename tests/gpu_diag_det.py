"""dev diagnostic: is the 17-stream forward bit-reproducible?  Runs the same sweep three times in one process and compares the
block buffers and the bottlenecks of every layer bit for bit against the first run."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from helpers import orc
import synthetic
from trainer import Trainer
import models

tr = Trainer('reinforcement', 0.5, False, None, False)
sd = synthetic.make_state_dict(orc.state_layout(1), 0)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
tr.model.gnum_rotations = tr.model.snum_rotations = 16
depth, masks = synthetic.heightmap_scene(0)
hm = tr._heightmaps_to_device(depth, depth * masks[0])
names = ["x1", "x2", "x3", "x4"] + ["bt%d_%d" % (b + 1, i + 1) for b in range(4) for i in range((6, 12, 24, 16)[b])]
ref = None
for it in range(3):
    q = tr.model.run(0, list(range(16)), 16, heightmaps=hm, mean=0.01, std=0.03, update_bn=False)
    torch.cuda.synchronize()
    eng = models._ENGINES[(0, 640, 1)]
    cur = {n: eng.debug_read(n).copy() for n in names}
    cur["q"] = q.reshape(-1).cpu().numpy()
    cur["fs"] = eng.debug_read("fs_bt1_1").view(np.float64).copy()
    if ref is None:
        ref = cur
        continue
    bad = []
    for n in names + ["q"]:
        d = cur[n] != ref[n]
        if d.any():
            a, b = cur[n][d].astype(np.float64), ref[n][d].astype(np.float64)
            bad.append("%s: %d differ, max rel %.2e" % (n, int(d.sum()), float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))))
    print("run %d: %d of %d buffers differ" % (it, len(bad), len(names) + 1))
    NS, H, HWp = eng.max_streams, eng.H, eng.HWp
    fa, fb = cur["fs"].reshape(2, NS, 128), ref["fs"].reshape(2, NS, 128)
    print("    fp64 sums of bt1_1: differing (stream, channel) entries: sum", int((fa[0] != fb[0]).sum()), "sumsq", int((fa[1] != fb[1]).sum()),
          "max rel (sumsq)", float(np.max(np.abs(fa[1] - fb[1]) / np.maximum(np.abs(fb[1]), 1e-300))), "max |dsum| / sqrt(n sumsq)", float(np.max(np.abs(fa[0] - fb[0]) / np.sqrt(25600 * np.maximum(fb[1], 1e-300)))), "streams with a differing sumsq:", np.nonzero((fa[1] != fb[1]).any(axis=1))[0].tolist())
    d1 = (cur["x1"] != ref["x1"]).reshape(NS, HWp[2], 256)
    print("    x1 differing elements per 32-channel slice:", d1.reshape(NS, HWp[2], 8, 32).sum(axis=(0, 1, 3)).tolist())
    print("    x1 slice 2 (layer 1 output) differing per stream:", d1[:, :, 64:96].sum(axis=(1, 2)).tolist())
    db = (cur["bt1_2"] != ref["bt1_2"]).reshape(NS, HWp[2], 128)
    print("    bt1_2 differing per stream:", db.sum(axis=(1, 2)).tolist())
    rows = d1[0, :25600, 64:96].any(axis=1).reshape(160, 160)
    print("    stream 0, layer-1 output: differing pixel rows (y):", np.nonzero(rows.any(axis=1))[0][:40].tolist(), "cols:", np.nonzero(rows.any(axis=0))[0][:40].tolist())
    for l in bad[:6]:
        print("   ", l)
