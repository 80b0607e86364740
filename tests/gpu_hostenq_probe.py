import sys, time, numpy as np, torch
sys.path.insert(0, 'smg-multimodal-grasping_amd'); sys.path.insert(0, '.')
import synthetic, smg_hip, models
from trainer import Trainer
from oracle import affordance as orc
tr = Trainer('reinforcement', 0.5, False, None, False)
sd = synthetic.make_state_dict(orc.state_layout(1), 0)
tr.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
tr.model.gnum_rotations = tr.model.snum_rotations = 16
depth, masks = synthetic.heightmap_scene(0)
E = smg_hip.Engine
tim = {}
def wrap(name):
    f = getattr(E, name)
    def g(self, *a, **k):
        t = time.perf_counter(); r = f(self, *a, **k); tim[name] = tim.get(name, 0.0) + time.perf_counter() - t; return r
    setattr(E, name, g)
for n in ('forward', 'loss', 'backward'): wrap(n)
adam0 = smg_hip.adam_step
def adam(*a, **k):
    t = time.perf_counter(); r = adam0(*a, **k); tim['adam'] = tim.get('adam', 0.0) + time.perf_counter() - t; return r
smg_hip.adam_step = adam
import trainer as T; T.smg_hip.adam_step = adam
for i in range(4): tr.backprop(depth, 'grasp', (0, i), (0, 0), (0, 0), (0, 0), 0.5, masks.copy(), None, None, None)
torch.cuda.synchronize(); tim.clear()
N = 20; t0 = time.perf_counter()
for i in range(N): tr.backprop(depth, 'grasp', (0, i % 16), (0, 0), (0, 0), (0, 0), 0.5, masks.copy(), None, None, None)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
print('step %.3f ms; host in engine calls:' % tot, {k: round(v / N * 1e3, 3) for k, v in tim.items()}, 'sum %.3f' % (sum(tim.values()) / N * 1e3))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for i in range(N): tr.backprop(depth, 'grasp', (0, i % 16), (0, 0), (0, 0), (0, 0), 0.5, masks.copy(), None, None, None)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3500])
