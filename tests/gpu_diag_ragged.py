"""dev diagnostic: per-tensor gradient errors at S=704 (ragged planes, dense 3x3 Q map) vs the fp64 oracle."""
import copy, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import MEAN, STD, oracle_net, orc, product_net
import synthetic
style, rot = 0, 5
mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 240
on = oracle_net(2)
depth, masks = synthetic.heightmap_scene(8, size=size, n_boxes=8)
x = orc.preprocess(depth, [MEAN] * 3, [STD] * 3)
mx = orc.preprocess(depth * masks[0], [MEAN] * 3, [STD] * 3)
OH = x.shape[-1] // 32 - 19
wq = torch.from_numpy(synthetic.uniform(3, "ragged/wq", OH * OH, -1.0, 1.0).astype(np.float32)).reshape(1, 1, OH, OH)
if mode == "single":
    wq = torch.zeros(1, 1, OH, OH); wq[0, 0, 0, 0] = 1.0
rx = orc.rotate(x, rot, 16)
o64 = copy.deepcopy(on).double()
trunk = getattr(o64, orc.STYLE_TRUNK[style]).features
head = getattr(o64, orc.STYLE_HEAD[style])
q64 = head(torch.cat((trunk(rx.double()), trunk(mx.double())), 1))
(q64 * wq.double()).sum().backward()
g64 = {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}
on.zero_grad()
qo = orc.forward(on, x, mx, style, False, rot)
(qo * wq).sum().backward()
po = dict(on.named_parameters())
net = product_net(2)
net.zero_grad()
qp = net.forward(x, mx, style, False, rot)
(qp * wq.cuda()).sum().backward()
print("q", qp.detach().cpu().numpy().ravel(), q64.detach().numpy().ravel())
rows = []
for name, p in net.named_parameters():
    if name not in g64: continue
    t = g64[name].numpy(); e = np.sqrt(((p.grad.cpu().double().numpy() - t) ** 2).sum()); n = np.sqrt((t * t).sum())
    eo = np.sqrt(((po[name].grad.double().numpy() - t) ** 2).sum())
    rows.append((e / max(n, 1e-30), eo / max(n, 1e-30), name, n))
rows.sort(reverse=True)
print("S", x.shape[-1], "median rel err: product %.3e oracle-fp32 %.3e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
for r in rows[:12]: print("prod %.3e  orc32 %.3e  %-66s |g| %.3e" % r)
for r in rows:
    if "graspnet_val" in r[2]: print("prod %.3e  orc32 %.3e  %-66s |g| %.3e" % r)
