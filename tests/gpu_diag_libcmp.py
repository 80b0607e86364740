"""dev diagnostic (not a test): one forward + backward of the headline batch under two builds of the library
(SMG_HIP_LIB), printed as checksums - a change that must not alter a single bit (instruction selection, scheduling) is
checked by comparing the two lines.  usage: python tests/gpu_diag_libcmp.py libA.so libB.so"""
import hashlib, os, subprocess, sys

CHILD = r'''
import os, sys, hashlib
sys.path.insert(0, "tests"); sys.path.insert(0, "smg-multimodal-grasping_amd")
import numpy as np, torch
import synthetic
from trainer import Trainer
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer('reinforcement', 0.5, False, None, False)
sys.path.insert(0, ".")
import bench
sd = synthetic.make_state_dict(bench.layout_names(), 0)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
tr.model.gnum_rotations = tr.model.snum_rotations = 16
depth, masks = synthetic.heightmap_scene(0)
labels = synthetic.uniform(0, "bench/labels", 16, 0.0, 1.5)
q = tr.forward(depth, depth * masks[0], 0, True, False, -1)
loss = tr.train_batch(depth, depth * masks[0], 0, list(range(16)), labels)
torch.cuda.synchronize()
g = tr.model.flat_grads().detach().cpu().numpy()
p = torch.cat([t.detach().reshape(-1).cpu() for t in tr.model.parameters()]).numpy()
print("RESULT q", hashlib.sha1(np.asarray(q).tobytes()).hexdigest()[:12], "grads", hashlib.sha1(g.tobytes()).hexdigest()[:12],
      "params", hashlib.sha1(p.tobytes()).hexdigest()[:12], "q0 %.9g" % float(np.asarray(q).reshape(-1)[0]), "gnorm %.9g" % float(np.sqrt((g.astype(np.float64) ** 2).sum())))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "-":
        env["SMG_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(lib, lines[0] if lines else out.stderr[-800:])
