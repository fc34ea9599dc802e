"""Device time of the training step when the host waits after every launch against launches
issued back to back:   python tools/sync_vs_async.py <workload> <f32|bf16>
(finding 54: does a graph with a side branch depend on the host ringing a doorbell?)"""
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
ctx = plan.ctx
torch.cuda.synchronize()


def timed(n, sync):
    tot = 0.0
    e0, e1 = ctx.event(), ctx.event()
    old = ctx.stream
    if not sync:
        ctx.set_stream(plan.stream); ctx.record(e0); ctx.set_stream(old)
        for _ in range(n):
            plan.run()
        ctx.set_stream(plan.stream); ctx.record(e1); ctx.set_stream(old)
        torch.cuda.synchronize()
        return ctx.elapsed_ms(e0, e1) / n
    for _ in range(n):
        ctx.set_stream(plan.stream); ctx.record(e0); ctx.set_stream(old)
        plan.run()
        ctx.set_stream(plan.stream); ctx.record(e1); ctx.set_stream(old)
        torch.cuda.synchronize()
        tot += ctx.elapsed_ms(e0, e1)
    return tot / n


timed(5, False)
a = timed(40, False)
s = timed(40, True)
a2 = timed(40, False)
print("%s %s: back to back %.4f / %.4f ms per step, host waiting after every step %.4f ms" % (wl, mode, a, a2, s))
