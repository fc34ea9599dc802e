"""build unet3d_lite at its native size, run training steps, report time"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import nets
from elektronn2_amd import neuromancer as nm
nm.model_manager.reset()
np.random.seed(0)
m = nets.unet3d_lite()
print("out shape", m.prediction_node.shape, "params", sum(int(np.prod(p.get_value().shape)) for p in m.loss_node.all_trainable_params.values()))
m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
rng = np.random.RandomState(0)
x = rng.rand(1, 1, 22, 140, 140).astype(np.float32)
osp = m.prediction_node.shape.spatial_shape
t = rng.randint(0, 2, [1, 1] + list(osp)).astype(np.float32)
for i in range(6):
    t0 = time.time()
    loss, dt, _ = m.trainingstep(x, t, optimiser='Adam')
    torch.cuda.synchronize()
    print(i, float(loss), "wall %.1f ms" % ((time.time() - t0) * 1e3), "device %.3f ms" % (m.optimisers['Adam'].step.func.last_device_time * 1e3))
