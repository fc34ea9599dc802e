"""The weight repack alone (debug build: E2_PACK_GRID = blocks per job):
    E2HIP_LIB=.../build/dbg/libe2hip.so E2_PACK_GRID=1024 python tools/pack_bench.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from elektronn2_amd import nets

wl = sys.argv[1] if len(sys.argv) > 1 else "full185"
builder, sp, _ = bench.WORKLOADS[wl]
np.random.seed(1)
model = getattr(nets, builder)((None, 1) + sp)
osp = tuple(model.prediction_node.shape.spatial_shape)
model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
opt = model.optimisers['Adam']
opt.step.compile()
plan = opt.step.func
x = torch.rand((1, 1) + sp, device="cuda"); t = torch.zeros((1, 1) + osp, device="cuda")
plan.set_inputs([x, t])
ctx = plan.ctx
old = ctx.stream
ctx.set_stream(plan.stream)
with torch.cuda.stream(plan.stream):
    for _ in range(5):
        ctx.conv3d_pack_multi(*plan._pack_dev)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(50):
        ctx.conv3d_pack_multi(*plan._pack_dev)
    ctx.record(e1)
torch.cuda.synchronize()
ctx.set_stream(old)
print("%s: repack of %d images, grid %s: %.1f us per launch (back to back)"
      % (wl, plan._pack_dev[1], os.environ.get("E2_PACK_GRID", "512"), ctx.elapsed_ms(e0, e1) / 50 * 1e3))
