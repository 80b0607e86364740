"""dev probe: how much do two independent forward sweeps gain from running concurrently on two HIP streams?
(reinforcement_net and reactive_net have different engines, so they can run side by side.)"""
import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import product_net, scene_tensors
a = product_net(0, 1); b = product_net(1, 3)
x, mx = scene_tensors(0, [0])
x = x.cuda(); mx = mx.cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(net, stream, n):
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(n):
            net.forward(x, mx, 0, True, -1)
N = 20
for net, st in ((a, sa), (b, sb)): run(net, st, 3)
torch.cuda.synchronize()
res = {}
for name, net, st in (("A", a, sa), ("B", b, sb)):
    t = time.perf_counter(); run(net, st, N); torch.cuda.synchronize(); res[name] = (time.perf_counter() - t) / N * 1e3
t = time.perf_counter()
ta = threading.Thread(target=run, args=(a, sa, N)); tb = threading.Thread(target=run, args=(b, sb, N))
ta.start(); tb.start(); ta.join(); tb.join(); torch.cuda.synchronize()
both = (time.perf_counter() - t) / N * 1e3
print("sweep A %.2f ms, sweep B %.2f ms, concurrent pair %.2f ms -> %.1f%% of the serial sum" % (res["A"], res["B"], both, 100 * both / (res["A"] + res["B"])))
