"""How far are the fp32 oracle and the HIP path from an fp64 evaluation of the same sweep?"""
import copy
import numpy as np
import torch
from helpers import oracle_net, orc, product_net, scene_tensors

for seed in (0, 1):
    on = oracle_net(seed)
    x, mx = scene_tensors(seed, [seed % 8])
    o64 = copy.deepcopy(on).double()
    pn = product_net(seed)
    qp = np.asarray([float(t) for t in pn.forward(x, mx, 0, True, -1)])
    q32, q64 = [], []
    with torch.no_grad():
        for r in range(16):
            rx = orc.rotate(x, r, 16)
            f = on.grasp_depth_trunk.features
            q32.append(float(on.graspnet_val(torch.cat((f(rx), f(mx)), 1))))
            f = o64.grasp_depth_trunk.features
            q64.append(float(o64.graspnet_val(torch.cat((f(rx.double()), f(mx.double())), 1))))
    q32, q64 = np.asarray(q32), np.asarray(q64)
    print("seed", seed, "max|Q|", np.abs(q64).max())
    print("  oracle32 - fp64 :", np.array2string(q32 - q64, precision=2))
    print("  product  - fp64 :", np.array2string(qp - q64, precision=2))
    print("  max abs: oracle32 %.3e  product %.3e  product-vs-oracle32 %.3e" % (np.abs(q32 - q64).max(), np.abs(qp - q64).max(), np.abs(qp - q32).max()))
