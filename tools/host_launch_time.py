"""Host time of one graph launch of the training step against the step's time on the device:
    python tools/host_launch_time.py <workload> <f32|bf16>
(is the host, enqueueing a graph with parallel branches, the one the device waits for?)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
import elektronn2_amd
from elektronn2_amd import nets, neuromancer as nm

wl, mode = sys.argv[1], sys.argv[2]
builder, sp, _ = B.WORKLOADS[wl]
if mode == "bf16":
    elektronn2_amd.set_mfma_dtype("bf16")
np.random.seed(1)
m = getattr(nets, builder)((None, 1) + sp)
m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
osp = tuple(m.prediction_node.shape.spatial_shape)
x = np.random.rand(1, 1, *sp).astype(np.float32)
t = np.random.randint(0, 2, (1, 1) + osp).astype(np.float32)
for _ in range(4):
    m.trainingstep(x, t, optimiser='Adam')
plan = m.optimisers['Adam'].step.func
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(50):
    a = time.perf_counter()
    plan.run()
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s %s: host per launch median %.1f us (min %.1f, max %.1f); 50 launches issued in %.2f ms, done after %.2f ms"
      % (wl, mode, 1e6 * float(np.median(host)), 1e6 * min(host), 1e6 * max(host), 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
