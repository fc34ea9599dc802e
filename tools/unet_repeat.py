import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import neuromancer as nm, nets
nm.model_manager.reset()
np.random.seed(5)
model = nets.unet3d_lite()
rng = np.random.RandomState(6)
x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
t = rng.randint(0, 2, (1, 1, 10, 52, 52)).astype(np.float32)
names = list(model.loss_node.all_trainable_params.keys())
ref = None
for it in range(12):
    g = model.gradients(x, t)
    if ref is None:
        ref = g
        continue
    worst = []
    for i, n in enumerate(names):
        sc = np.abs(ref[i]).max() + 1e-30
        e = np.abs(g[i] - ref[i]).max() / sc
        if e > 2e-6:
            worst.append((n, float(e)))
    print(it, "tensors differing > 2e-6 from run 0:", worst)
