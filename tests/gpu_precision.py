"""Reduced-precision diagnostic (not a pytest file): 16-rotation Q sweeps and one training gradient with bf16 / fp16
MFMA operands against the default fp32-class path.  python tests/gpu_precision.py"""
import time

import numpy as np
import torch

from helpers import product_net, scene_tensors


def main():
    for seed in (0, 1, 2):
        net = product_net(seed)
        x, mx = scene_tensors(seed, [seed % 8])
        ref = None
        for prec in ("fp32", "bf16", "fp16"):
            net.set_precision(prec)
            with torch.no_grad():
                q = np.asarray([float(t) for t in net.forward(x, mx, 0, True, -1)])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                for _ in range(3):
                    net.forward(x, mx, 0, True, -1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            net.zero_grad()
            qp = net.forward(x, mx, 0, False, 5)
            (qp[0, 0, 0, 0] * 1.0).backward()
            g = net.flat_grads().double().cpu().numpy().copy()
            if ref is None:
                ref, gref = q, g
            top2 = np.sort(ref)[-2:]
            cos = float((g * gref).sum() / max(np.sqrt((g * g).sum() * (gref * gref).sum()), 1e-300))
            print("seed %d %-5s sweep %.2f ms  max|dq| %.3e (max|q| %.3f, top-2 margin %.3e)  argmax %d vs %d  grad cos %.6f  |g| ratio %.4f" % (
                seed, prec, ms, np.abs(q - ref).max(), np.abs(ref).max(), top2[1] - top2[0], int(q.argmax()), int(ref.argmax()), cos,
                np.sqrt((g * g).sum() / max((gref * gref).sum(), 1e-300))))


if __name__ == "__main__":
    main()
