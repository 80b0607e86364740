"""Gradient-noise diagnostic (not a pytest file): per-tensor error of the HIP path and of the PyTorch-CPU fp32 oracle
against an fp64 evaluation of the same train step, as quantiles.  python tests/gpu_gradnoise.py [style rot label]"""
import copy
import sys

import numpy as np
import torch

from helpers import oracle_net, orc, product_net, scene_tensors


def main(style=0, rot=3, label=0.4):
    on = oracle_net(0)
    x, mx = scene_tensors(0, [0])
    rx = orc.rotate(x, rot, 16)
    o64 = copy.deepcopy(on).double()
    trunk = getattr(o64, orc.STYLE_TRUNK[style]).features
    head = getattr(o64, orc.STYLE_HEAD[style])
    q64 = head(torch.cat((trunk(rx.double()), trunk(mx.double())), 1))
    orc.huber(q64[0, 0, 0, 0], label).sum().backward()
    g64 = {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}
    on.zero_grad()
    qo = orc.forward(on, x, mx, style, False, rot)
    orc.huber(qo[0, 0, 0, 0], label).sum().backward()
    net = product_net(0)
    net.zero_grad()
    qp = net.forward(x, mx, style, False, rot)
    d = qp[0, 0, 0, 0] - label
    (0.5 * d ** 2 if abs(float(d.detach())) < 1 else abs(d) - 0.5).backward()
    po = dict(on.named_parameters())
    rp, ro, names = [], [], []
    for name, p in net.named_parameters():
        if name not in g64:
            continue
        t = g64[name].numpy()
        nrm = max(np.sqrt((t * t).sum()), 1e-30)
        rp.append(np.sqrt(((p.grad.cpu().double().numpy() - t) ** 2).sum()) / nrm)
        ro.append(np.sqrt(((po[name].grad.double().numpy() - t) ** 2).sum()) / nrm)
        names.append(name)
    rp, ro = np.asarray(rp), np.asarray(ro)
    print("q: fp64 %.8f  oracle %.8f  hip %.8f" % (float(q64), float(qo), float(qp)))
    for q in (10, 50, 90, 100):
        print("  percentile %3d: hip %.3e  oracle-fp32 %.3e" % (q, np.percentile(rp, q), np.percentile(ro, q)))
    ratio = rp / np.maximum(ro, 1e-12)
    print("  per-tensor ratio hip/oracle: median %.2f  p90 %.2f  max %.2f (%s)" % (np.median(ratio), np.percentile(ratio, 90), ratio.max(), names[int(ratio.argmax())]))
    order = np.argsort(-rp)[:8]
    for i in order:
        print("   worst %-70s hip %.3e oracle %.3e" % (names[i], rp[i], ro[i]))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(*(int(a[0]), int(a[1]), float(a[2])) if len(a) == 3 else ())
