import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import neuromancer as nm, nets
from oracle import e2_oracle as O
which = sys.argv[1]
nm.model_manager.reset()
np.random.seed(5)
rng = np.random.RandomState(6)
if which == 'unet':
    model = nets.unet3d_lite()
    x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
    t = rng.randint(0, 2, (1, 1, 10, 52, 52)).astype(np.float32)
elif which == 'unet_small':
    model = nets.unet3d_lite((None, 1, 22, 100, 100))
    osp = model.prediction_node.shape.spatial_shape
    x = rng.rand(1, 1, 22, 100, 100).astype(np.float32)
    t = rng.randint(0, 2, [1, 1] + list(osp)).astype(np.float32)
else:
    model = nets.neuro3d_lite((None, 1, 23, 183, 183), params=O.init_net(O.NEURO3D_LITE, 1, seed=1))
    x = rng.rand(1, 1, 23, 183, 183).astype(np.float32)
    t = rng.randint(0, 2, (1, 1, 10, 37, 37)).astype(np.float32)
names = list(model.loss_node.all_trainable_params.keys())
ref = None
for it in range(7):
    g = model.gradients(x, t)
    L = float(model.loss(x, t))
    if ref is None:
        ref = g
        print(it, "loss", L)
        continue
    errs = [float(np.abs(g[i] - ref[i]).max() / (np.abs(ref[i]).max() + 1e-30)) for i in range(len(names))]
    k = int(np.argmax(errs))
    print(it, "loss %.8f" % L, "max rel diff vs run0: %.3e (%s)" % (errs[k], names[k]), "n>1e-5:", sum(e > 1e-5 for e in errs))
