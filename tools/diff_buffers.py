import sys, os
os.environ['E2_NO_GRAPH'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import neuromancer as nm, nets
nm.model_manager.reset()
np.random.seed(5)
rng = np.random.RandomState(6)
model = nets.unet3d_lite()
x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
t = rng.randint(0, 2, (1, 1, 10, 52, 52)).astype(np.float32)
model._grad_func(x, t)
plan = model._grad_func.func
print("plan", type(plan).__name__)
def snap():
    d = {}
    for n in plan.nodes:
        for kind, dic in (('out', plan.out), ('grad', plan.grad)):
            b = dic.get(n)
            if b is not None:
                d[(kind, n.name)] = b.detach().cpu().numpy().copy()
        for key in ('dy',):
            b = plan.scratch.get((n, key))
            if b is not None:
                d[(key, n.name)] = b.detach().cpu().numpy().copy()
    d[('G', '')] = model.G.detach().cpu().numpy().copy()
    return d
snaps = []
for it in range(4):
    model._grad_func(x, t)
    snaps.append(snap())
for it in range(1, 4):
    bad = []
    for k in snaps[0]:
        a, b = snaps[0][k], snaps[it][k]
        sc = np.abs(a).max() + 1e-30
        e = float(np.abs(a - b).max() / sc)
        if e > 2e-6:
            bad.append((k, "%.2e" % e))
    print(it, "buffers differing from run 0:", bad[:40])
