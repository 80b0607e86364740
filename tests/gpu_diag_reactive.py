"""dev diagnostic: error of the stem gradients (norm0, conv0) of the reactive train step against the fp64 oracle."""
import copy, sys
import numpy as np, torch
from helpers import oracle_net, orc, product_net, scene_tensors

def main():
    for out_ch, R, rot in ((3, 1, 0), (1, 16, 0), (1, 16, 3)):
        on = oracle_net(0, out_ch=out_ch, R=R)
        x, mx = scene_tensors(0, [0])
        o64 = copy.deepcopy(on).double(); o64.zero_grad()
        trunk, head = getattr(o64, orc.STYLE_TRUNK[0]).features, getattr(o64, orc.STYLE_HEAD[0])
        rx = orc.rotate(x, rot, R)
        q64 = head(torch.cat((trunk(rx.double()), trunk(mx.double())), 1))
        if out_ch == 3:
            lab = torch.ones((1, 1, 1), dtype=torch.long)
            torch.nn.functional.nll_loss(torch.log_softmax(q64[0].view(1, 3, 1, 1), dim=1), lab, weight=torch.tensor([1.0, 1.0, 0.0], dtype=torch.float64)).sum().backward()
        else:
            orc.huber(q64[0, 0, 0, 0], 0.4).sum().backward()
        g64 = {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}
        net = product_net(0, out_ch=out_ch, R=R); net.zero_grad()
        q = net.forward(x, mx, 0, False, rot)
        if out_ch == 3:
            w = torch.tensor([1.0, 1.0, 0.0], device=q.device); label = torch.ones((1, 1, 1), dtype=torch.long, device=q.device)
            torch.nn.functional.nll_loss(torch.log_softmax(q[0].view(1, 3, 1, 1), dim=1), label, weight=w).sum().backward()
        else:
            d = q[0, 0, 0, 0] - 0.4; (0.5 * d ** 2 if abs(float(d.detach())) < 1 else abs(d) - 0.5).backward()
        pp = dict(net.named_parameters())
        for k in ("grasp_depth_trunk.features.conv0.weight", "grasp_depth_trunk.features.norm0.weight", "grasp_depth_trunk.features.norm0.bias",
                  "grasp_depth_trunk.features.denseblock1.denselayer1.norm1.bias", "grasp_depth_trunk.features.denseblock1.denselayer1.conv1.weight"):
            t = g64[k].numpy(); g = pp[k].grad.cpu().double().numpy()
            print("out_ch %d rot %d %-62s rel err %.3e  |g| %.3e" % (out_ch, rot, k, np.sqrt(((g - t) ** 2).sum() / (t * t).sum()), np.sqrt((t * t).sum())))
        if out_ch == 3:
            t = g64["grasp_depth_trunk.features.norm0.bias"].numpy(); g = pp["grasp_depth_trunk.features.norm0.bias"].grad.cpu().double().numpy()
            bad = np.argsort(-np.abs(g - t))[:6]
            print("worst channels", bad, (g - t)[bad], t[bad])

if __name__ == "__main__":
    main()
