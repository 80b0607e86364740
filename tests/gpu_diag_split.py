"""dev diagnostic (not a test): the same 17-stream forward + backward through two builds of the library (SMG_HIP_LIB), per-layer
comparison of the block buffers, the bottlenecks and the gradients - finds the first layer where an arithmetic variant deviates.
  python tests/gpu_diag_split.py dump out.npz        (run once per build)
  python tests/gpu_diag_split.py cmp a.npz b.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def dump(path, nrot):
    import torch
    from helpers import orc
    import synthetic
    from trainer import Trainer
    import models
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = 16
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(0)
    rots = list(range(nrot))
    labels = synthetic.uniform(0, "bench/labels", 16, 0.0, 1.5)[:nrot]
    loss, q = tr.train_batch(depth, depth * masks[0], 0, rots, labels, return_q=True)
    torch.cuda.synchronize()
    eng = models._ENGINES[(0, 640, 1)]
    out = {"q": q.reshape(-1).cpu().numpy(), "loss": loss.cpu().numpy()}
    NS, H, HWp = eng.max_streams, eng.H, eng.HWp
    ns = nrot + 1
    Ct = (256, 512, 1024, 1024)
    for b in range(4):
        x = eng.debug_read("x%d" % (b + 1)).reshape(NS, HWp[2 + b], Ct[b])[:ns, :H[2 + b] * H[2 + b]].astype(np.float64)
        out["x%d_slice_norm" % (b + 1)] = np.sqrt((x * x).sum(axis=(0, 1)).reshape(-1, 32).sum(axis=1))      # per 32-channel slice
        out["x%d_probe" % (b + 1)] = x[:, ::97, ::7].ravel()[:4096]
        g = eng.debug_read("g%d" % (b + 1)).reshape(NS, HWp[2 + b], Ct[b])[:ns, :H[2 + b] * H[2 + b]].astype(np.float64)
        out["g%d_slice_norm" % (b + 1)] = np.sqrt((g * g).sum(axis=(0, 1)).reshape(-1, 32).sum(axis=1))
        out["g%d_probe" % (b + 1)] = g[:, ::97, ::7].ravel()[:4096]
        for i in (1, (6, 12, 24, 16)[b]):
            bt = eng.debug_read("bt%d_%d" % (b + 1, i)).reshape(NS, HWp[2 + b], 128)[:ns, :H[2 + b] * H[2 + b]].astype(np.float64)
            out["bt%d_%d_probe" % (b + 1, i)] = bt[:, ::89, ::5].ravel()[:4096]
    names, norms, probes = [], [], []
    for n_, p in tr.model.named_parameters():
        if p.grad is not None:
            gg = p.grad.double().cpu().numpy().ravel()
            names.append(n_); norms.append(np.sqrt((gg * gg).sum())); probes.append(gg[:: max(1, gg.size // 64)][:64])
    out["grad_names"] = np.asarray(names)
    out["grad_norms"] = np.asarray(norms)
    out["grad_probes"] = np.concatenate(probes)
    np.savez(path, **out)
    print("dumped", path, "q", out["q"][:4])


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    for k in A.files:
        if k == "grad_names":
            continue
        x, y = A[k].astype(np.float64), B[k].astype(np.float64)
        if k.endswith("slice_norm"):
            rel = np.abs(x - y) / np.maximum(np.abs(y), 1e-30)
            print("%-16s max rel diff %.3e at slice %d   [%s]" % (k, rel.max(), int(rel.argmax()), " ".join("%.1e" % v for v in rel[:40])))
        elif k == "grad_norms":
            rel = np.abs(x - y) / np.maximum(np.abs(y), 1e-30)
            order = np.argsort(-rel)[:12]
            for i in order:
                print("grad %-70s rel norm diff %.3e (|g| %.3e)" % (A["grad_names"][i], rel[i], y[i]))
            print("grad norms: median rel diff %.3e" % np.median(rel))
        else:
            d = np.sqrt(((x - y) ** 2).sum()) / max(np.sqrt((y * y).sum()), 1e-30)
            print("%-16s rel L2 diff %.3e" % (k, d))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    else:
        cmp(sys.argv[2], sys.argv[3])
