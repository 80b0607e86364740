"""Stage-by-stage GPU-vs-oracle diagnostic (not a pytest file): prints the error of
every intermediate buffer and every gradient tensor, never stops at the first one.
Run on the GPU box:  python tests/gpu_diag.py [fwd|bwd|all]  > gpurun_out/diag.txt"""
import sys
import time

import numpy as np
import torch

from helpers import MEAN, STD, nhwc_plane, oracle_net, orc, product_net, scene, scene_tensors

import models  # noqa: E402


def err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    return "max_abs %.3e  rel_l2 %.3e  ref_absmax %.3e  nan %d" % (
        d.max(), np.sqrt((d * d).sum()) / max(np.sqrt((b * b).sum()), 1e-30), np.abs(b).max(), int(np.isnan(a).sum()))


def main(what):
    torch.set_num_threads(max(1, torch.get_num_threads()))
    R, rot = 16, 3
    on = oracle_net(0)
    pn = product_net(0)
    x, mx = scene_tensors(0, [0])
    print("device", torch.cuda.get_device_name(0))

    # ---------------- forward, single sample (rot 3) --------------------------------
    t0 = time.time()
    qp = pn.forward(x, mx, 0, True, rot)
    torch.cuda.synchronize()
    print("product forward ok %.2fs q=%s" % (time.time() - t0, qp.cpu().numpy().ravel()))
    eng = models._ENGINES[(0, 640, 1)]
    NS = eng.max_streams
    H, HWp = eng.H, eng.HWp
    feats = on.grasp_depth_trunk.features

    def capture(inp):
        import copy
        f2 = copy.deepcopy(feats)
        out = {}
        for name in ("conv0", "pool0", "denseblock1", "transition1", "denseblock2", "transition2", "denseblock3",
                     "transition3", "denseblock4", "norm5"):
            getattr(f2, name).register_forward_hook(lambda m, i, o, name=name: out.__setitem__(name, o.detach().numpy()[0]))
        f2.denseblock1.denselayer1.conv1.register_forward_hook(lambda m, i, o: out.__setitem__("b1l1c1", o.detach().numpy()[0]))
        f2.denseblock1.denselayer1.conv2.register_forward_hook(lambda m, i, o: out.__setitem__("b1l1c2", o.detach().numpy()[0]))
        with torch.no_grad():
            f2(inp)
        return out

    with torch.no_grad():
        rx = orc.rotate(x, rot, R)
    stages = capture(rx)
    stages_m = capture(mx)
    with torch.no_grad():
        qo = orc.forward(on, x, mx, 0, True, rot)
    print("oracle q", float(qo), " product q", float(qp), " abs diff %.3e" % abs(float(qo) - float(qp)))

    img = eng.debug_read("img")
    a = nhwc_plane(img, NS, HWp[0], 4, 640, 640, 0)
    print("img rot  (bit-exact expected):", err(a[:3], rx.numpy()[0]), " mismatches", int((a[:3] != rx.numpy()[0]).sum()), " ch3 absmax", np.abs(a[3]).max())
    a = nhwc_plane(img, NS, HWp[0], 4, 640, 640, 1)
    print("img mask (bit-exact expected):", err(a[:3], mx.numpy()[0]), " mismatches", int((a[:3] != mx.numpy()[0]).sum()))
    st = nhwc_plane(eng.debug_read("stem"), NS, HWp[1], 64, H[1], H[1], 0)
    print("stem conv0      :", err(st, stages["conv0"]))
    x1 = nhwc_plane(eng.debug_read("x1"), NS, HWp[2], 256, H[2], H[2], 0)
    print("pool0           :", err(x1[:64], stages["pool0"]))
    bt = nhwc_plane(eng.debug_read("bt1_1"), NS, HWp[2], 128, H[2], H[2], 0)
    print("b1 l1 conv1     :", err(bt, stages["b1l1c1"]))
    print("b1 l1 conv2     :", err(x1[64:96], stages["b1l1c2"]))
    print("denseblock1     :", err(x1, stages["denseblock1"]))
    for c0 in range(64, 256, 32):
        print("   block1 ch %3d..:" % c0, err(x1[c0:c0 + 32], stages["denseblock1"][c0:c0 + 32]))
    x2 = nhwc_plane(eng.debug_read("x2"), NS, HWp[3], 512, H[3], H[3], 0)
    print("transition1     :", err(x2[:128], stages["transition1"]))
    print("denseblock2     :", err(x2, stages["denseblock2"]))
    x3 = nhwc_plane(eng.debug_read("x3"), NS, HWp[4], 1024, H[4], H[4], 0)
    print("transition2     :", err(x3[:256], stages["transition2"]))
    print("denseblock3     :", err(x3, stages["denseblock3"]))
    x4 = nhwc_plane(eng.debug_read("x4"), NS, HWp[5], 1024, H[5], H[5], 0)
    print("transition3     :", err(x4[:512], stages["transition3"]))
    print("denseblock4     :", err(x4, stages["denseblock4"]))
    ft = eng.debug_read("feat").reshape(eng.max_pairs, HWp[5], 2048)[0, :400].reshape(20, 20, 2048).transpose(2, 0, 1)
    print("norm5 (rot half):", err(ft[:1024], stages["norm5"]))
    print("norm5 (mask half):", err(ft[1024:], stages_m["norm5"]))
    sys.stdout.flush()

    # ---------------- sweep -----------------------------------------------------------
    t0 = time.time()
    ql = pn.forward(x, mx, 0, True, -1)
    torch.cuda.synchronize()
    print("product 16-rot sweep %.3fs" % (time.time() - t0))
    qs = np.asarray([float(t) for t in ql])
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))
    print("sweep vs golden G4 seed0 style0:", err(qs, g["g4_s0_q0"]), " argmax", int(qs.argmax()), int(g["g4_s0_q0"].argmax()))
    print("  product:", np.array2string(qs, precision=5))
    print("  golden :", np.array2string(g["g4_s0_q0"], precision=5))
    sys.stdout.flush()
    if what == "fwd":
        return

    # ---------------- backward --------------------------------------------------------
    for style, rot, label in ((0, 3, 0.4), (2, 0, -3.0)):
        on = oracle_net(0)
        pn = product_net(0)
        on.zero_grad()
        qo = orc.forward(on, x, mx, style, False, rot)
        loss_o = orc.huber(qo[0, 0, 0, 0], label).sum()
        loss_o.backward()
        pn.zero_grad()
        qp = pn.forward(x, mx, style, False, rot)
        d = qp[0, 0, 0, 0] - label
        loss_p = 0.5 * d ** 2 if abs(float(d)) < 1 else abs(d) - 0.5
        loss_p.backward()
        torch.cuda.synchronize()
        print("style %d: oracle q %.6f loss %.6f | product q %.6f loss %.6f" % (style, float(qo), float(loss_o), float(qp), float(loss_p)))
        po = dict(on.named_parameters())
        rows = []
        for name, p in pn.named_parameters():
            go = po[name].grad
            if go is None:
                if p.grad is not None:
                    print("  UNEXPECTED grad on", name)
                continue
            if p.grad is None:
                print("  MISSING grad on", name)
                continue
            a, b = p.grad.cpu().numpy().astype(np.float64), go.numpy().astype(np.float64)
            rel = np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b * b).sum()), 1e-30)
            rows.append((name, rel, np.sqrt((b * b).sum())))
        bad = [r for r in rows if not (r[1] < 2e-3)]
        print("  tensors compared %d, rel_l2 >= 2e-3: %d, worst %.3e, median %.3e" % (
            len(rows), len(bad), max(r[1] for r in rows), float(np.median([r[1] for r in rows]))))
        for r in rows[::-1][:12]:
            print("   (tail) %-70s rel %.3e  |g| %.3e" % r)
        for r in bad[::-1][:40]:
            print("   BAD    %-70s rel %.3e  |g| %.3e" % r)
        sys.stdout.flush()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "all")
