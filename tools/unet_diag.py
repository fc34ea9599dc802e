import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from elektronn2_amd import neuromancer as nm, nets
from test_model_gpu import torch_mirror
nm.model_manager.reset()
np.random.seed(5)
model = nets.unet3d_lite()
rng = np.random.RandomState(6)
x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
t = rng.randint(0, 2, (1, 1, 10, 52, 52)).astype(np.float32)
torch.set_num_threads(16)
t0 = time.time(); L64, G64 = torch_mirror(model, x, t, torch.float64); print("f64 mirror %.1f s" % (time.time() - t0))
t0 = time.time(); L32, G32 = torch_mirror(model, x, t, torch.float32); print("f32 mirror %.1f s" % (time.time() - t0))
Lg = float(model.loss(x, t)); g = model.gradients(x, t)
print("loss f64 %.8f f32 %.8f gpu %.8f" % (L64, L32, Lg))
names = list(model.loss_node.all_trainable_params.keys())
for i, n in enumerate(names):
    ref = G64[n]; sc = np.abs(ref).max() + 1e-30
    print("%-12s max|g| %.3e  gpu-vs-f64 %.2e  torch32-vs-f64 %.2e" % (n, sc, np.abs(g[i] - ref).max() / sc, np.abs(G32[n] - ref).max() / sc))
