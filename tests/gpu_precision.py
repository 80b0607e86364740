"""16-bit storage diagnostic (not a pytest file): BASELINE.json configs 3 (bf16) and 5 (fp16) against the default
fp32-class path - Q error, argmax, per-head gradient cosine / norm ratio, step time.  python tests/gpu_precision.py"""
import sys
import time

import numpy as np
import torch

from helpers import orc
import synthetic


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def cos(a, b):
    return float((a * b).sum() / max(np.sqrt((a * a).sum() * (b * b).sum()), 1e-300))


def main():
    from trainer import Trainer
    import smg_hip
    R = 16
    tr = Trainer('reinforcement', 0.5, False, None, False)
    sd = synthetic.make_state_dict(orc.state_layout(1), 0)
    tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    tr.model.gnum_rotations = tr.model.snum_rotations = R
    tr.optimizer.lr = 0.0
    depth, masks = synthetic.heightmap_scene(0)
    rots = list(range(R))
    labels = synthetic.uniform(0, "bench/labels", R, 0.0, 1.5)
    md, md2 = depth * masks[0], depth * (masks[1] + masks[2])
    which = sys.argv[1:] or ["c3", "c5"]
    if "c3" in which:
        res = {}
        for prec in ("fp32", "bf16", "fp16"):
            tr.model.set_precision(prec)
            out = []
            for style, m, rs, lab in ((0, md, rots, labels), (1, md, rots, labels), (2, md2, [0], labels[:1])):
                loss, q = tr.train_batch(depth, m, style, rs, lab, return_q=True)
                t0, n0 = smg_hip.trunk_range(1, (1, 0, 2)[style])
                h0, hn = smg_hip.head_range(1, (1, 0, 0)[style])
                g = tr.model.flat_grads().double().cpu().numpy()
                out.append((q.reshape(-1).cpu().numpy().astype(np.float64), np.concatenate([g[t0:t0 + n0], g[h0:h0 + hn]]), loss.cpu().numpy()))

            def three():
                tr.train_batch(depth, md, 0, rots, labels); tr.train_batch(depth, md, 1, rots, labels); tr.train_batch(depth, md2, 2, [0], labels[:1])
            res[prec] = (out, timed(three))
        ref = res["fp32"][0]
        for prec in ("bf16", "fp16"):
            out, ms = res[prec]
            for style in range(3):
                q, g, loss = out[style]; qr, gr, lr_ = ref[style]
                print("config3 %s style %d: %.2f ms (fp32-class %.2f) max|dq| %.4f of max|q| %.3f argmax %d vs %d finite %s | grad cos %.5f norm ratio %.4f | loss max diff %.4f" % (
                    prec, style, ms, res["fp32"][1], np.abs(q - qr).max(), np.abs(qr).max(), int(q.argmax()), int(qr.argmax()),
                    bool(np.isfinite(g).all() and np.isfinite(q).all()), cos(g, gr), np.sqrt((g * g).sum() / (gr * gr).sum()), np.abs(loss - lr_).max()))
    if "c5" in which:
        dbig, mbig = synthetic.heightmap_scene(4, size=640, n_boxes=8)
        tr.model.gnum_rotations = tr.model.snum_rotations = 32
        r5, l5 = [5, 6, 7, 8], [0.3, 1.9, 0.1, 0.7]
        res = {}
        for prec in ("fp32", "fp16", "bf16"):
            tr.model.set_precision(prec)
            loss, q = tr.train_batch(dbig, dbig * mbig[0], 0, r5, l5, return_q=True)
            g = tr.model.flat_grads().double().cpu().numpy().copy()
            ms = timed(lambda: tr.train_batch(dbig, dbig * mbig[0], 0, r5, l5))
            res[prec] = (q.reshape(4, -1).cpu().numpy().astype(np.float64), g, ms)
        qr, gr, msr = res["fp32"]
        for prec in ("fp16", "bf16"):
            q, g, ms = res[prec]
            print("config5 %s: %.2f ms (fp32-class %.2f) max|dq| %.4f of max|q| %.3f argmax agree %s finite %s | grad cos %.5f norm ratio %.4f" % (
                prec, ms, msr, np.abs(q - qr).max(), np.abs(qr).max(), [int(q[k].argmax()) == int(qr[k].argmax()) for k in range(4)],
                bool(np.isfinite(g).all()), cos(g, gr), np.sqrt((g * g).sum() / (gr * gr).sum())))


if __name__ == "__main__":
    main()
