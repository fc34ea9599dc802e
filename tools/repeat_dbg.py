import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import neuromancer as nm, nets
nm.model_manager.reset()
np.random.seed(5)
rng = np.random.RandomState(6)
model = nets.unet3d_lite()
x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
t = rng.randint(0, 2, (1, 1, 10, 52, 52)).astype(np.float32)
mode = sys.argv[1]
P0 = None
for it in range(6):
    if mode in ('both', 'grad'):
        g = model.gradients(x, t)
    if mode in ('both', 'loss'):
        L = float(model.loss(x, t))
    else:
        L = float('nan')
    P = model.P.detach().cpu().numpy().copy()
    if P0 is None:
        P0 = P
    print(it, "loss %.6f" % L, "params changed:", int((P != P0).sum()), "of", P.size,
          "first idx" , (np.nonzero(P != P0)[0][:5] if (P != P0).any() else None))
