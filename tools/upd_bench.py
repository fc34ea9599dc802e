"""The optimiser launch that writes the packed weight images (csrc/update_pack.hip) against the two
launches it replaces, isolated: python tools/upd_bench.py [lite183|full185|unet_lite140|unet132]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from elektronn2_amd import nets, neuromancer as nm

wl = sys.argv[1] if len(sys.argv) > 1 else "full185"
builder, sp, _ = B.WORKLOADS[wl]
np.random.seed(1)
m = getattr(nets, builder)((None, 1) + sp)
m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
opt = m.optimisers['Adam']
opt.step.compile()
plan = opt.step.func
osp = tuple(m.prediction_node.shape.spatial_shape)
x = np.random.rand(1, 1, *sp).astype(np.float32)
t = np.random.randint(0, 2, (1, 1) + osp).astype(np.float32)
m.trainingstep(x, t, optimiser='Adam')
ctx = plan.ctx
assert plan._upd is not None
G = m.G
G.normal_()
kw = opt._grad_scaling(plan)


def timeit(fn, n=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(n):
        fn()
    ctx.record(e1)
    torch.cuda.synchronize()
    return ctx.elapsed_ms(e0, e1) / n * 1e3


P = m.P[:m.n_train] if m.n_train < m.P.numel() else m.P
t_adam = timeit(lambda: ctx.adam_step(P, G, opt.momentum, opt.squared_accum, m.seg_off, m.seg_reg, opt._hyper, **kw))
t_pack = timeit(lambda: ctx.conv3d_pack_multi(*plan._pack_dev))
t_fused = timeit(lambda: ctx.adam_pack_step(m.P, G, opt.momentum, opt.squared_accum, plan._upd, opt._hyper, **kw))
print("%s: %d parameters, %d conv tensors, %d tiles, LDS %d B: adam %.1f us + pack %.1f us = %.1f us; fused %.1f us"
      % (wl, m.n_train, plan._upd['njobs'], plan._upd['ntiles'], plan._upd['lds'], t_adam, t_pack, t_adam + t_pack, t_fused))
