"""dev diagnostic: head backward intermediates at S=704 vs an fp64 evaluation."""
import copy, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import MEAN, STD, oracle_net, orc, product_net
import synthetic, models
size = int(sys.argv[1]) if len(sys.argv) > 1 else 240
style, rot = 0, 5
on = oracle_net(2)
depth, masks = synthetic.heightmap_scene(8, size=size, n_boxes=8)
x = orc.preprocess(depth, [MEAN] * 3, [STD] * 3)
mx = orc.preprocess(depth * masks[0], [MEAN] * 3, [STD] * 3)
S = x.shape[-1]; P = S // 32; OH = P - 19
wq = torch.zeros(1, 1, OH, OH); wq[0, 0, 0, 0] = 1.0
rx = orc.rotate(x, rot, 16)
o64 = copy.deepcopy(on).double()
trunk = getattr(o64, orc.STYLE_TRUNK[style]).features
head = getattr(o64, orc.STYLE_HEAD[style])
F = torch.cat((trunk(rx.double()), trunk(mx.double())), 1)
mods = list(head.children())
print([type(m).__name__ for m in mods])
t = F; keep = []
for m in mods:
    t = m(t); t.retain_grad(); keep.append(t)
q64 = t
(q64 * wq.double()).sum().backward()
bn1_out = keep[3]          # norm0, relu0, conv0, norm1, relu1, conv1
h1_pre = keep[2]           # conv0 output (input of norm1)
net = product_net(2)
net.zero_grad()
qp = net.forward(x, mx, style, False, rot)
(qp * wq.cuda()).sum().backward()
eng = models._ENGINES[(0, S, 1)]
HWp = (P * P + 63) // 64 * 64
h1 = eng.debug_read("h1").reshape(-1, HWp, 64)[0, :P * P].reshape(P, P, 64)
dh1 = eng.debug_read("dh1").reshape(-1, HWp, 64)[0, :P * P].reshape(P, P, 64)
ref_h1 = h1_pre.detach().numpy()[0].transpose(1, 2, 0)
ref_d = (keep[4].grad * (keep[4] > 0)).detach().numpy()[0].transpose(1, 2, 0)     # ReLU is in place: mask explicitly
print("h1 max abs err", np.abs(h1 - ref_h1).max(), "scale", np.abs(ref_h1).max())
err = np.abs(dh1 - ref_d)
print("dh1 max abs err", err.max(), "scale", np.abs(ref_d).max(), "sum dy ours", dh1.sum(), "ref", ref_d.sum())
bad = np.argwhere(err > 1e-4 * np.abs(ref_d).max())
print("bad elements", len(bad), "of", err.size)
ys = sorted(set(int(b[0]) for b in bad)); xs = sorted(set(int(b[1]) for b in bad))
print("rows with errors", ys, "cols", xs)
for b in bad[:10]: print(tuple(int(v) for v in b), dh1[tuple(b)], ref_d[tuple(b)], "h1", h1[tuple(b)], ref_h1[tuple(b)])
print("per-channel sum dy: worst", np.abs(dh1.sum((0, 1)) - ref_d.sum((0, 1))).max(), "ref |sum|", np.abs(ref_d.sum((0, 1))).max())
