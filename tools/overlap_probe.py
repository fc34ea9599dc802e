"""Would the weight repack hide under the fused first layer's forward if both were ONE launch?
Proxy: the two kernels back to back on one stream against the two on two streams with no
dependency between them (N iterations each):   python tools/overlap_probe.py <workload>"""
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
osp = tuple(m.prediction_node.shape.spatial_shape)
x = np.random.rand(1, 1, *sp).astype(np.float32)
t = np.random.randint(0, 2, (1, 1) + osp).astype(np.float32)
for _ in range(3):
    m.trainingstep(x, t, optimiser='Adam')
plan = m.optimisers['Adam'].step.func
ctx = plan.ctx
first = [n for n in plan.nodes if type(n).__name__ == 'Conv' and n._fused_first(plan)][0]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def pack():
    ctx.conv3d_pack_multi(*plan._pack_dev)


def fwd():
    first._plan_fwd(plan)


def on(stream, fn, n):
    old = ctx.stream
    ctx.set_stream(stream)
    with torch.cuda.stream(stream):
        for _ in range(n):
            fn()
    ctx.set_stream(old)


def timed(body):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    body()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


for _ in range(2):
    a = timed(lambda: on(s1, pack, N))
    b = timed(lambda: on(s1, fwd, N))
    ab = timed(lambda: on(s1, lambda: (pack(), fwd()), N))
    par = timed(lambda: (on(s1, pack, N), on(s2, fwd, N)))
    print("%s: repack %.1f us, first layer forward %.1f us, back to back %.1f us, on two streams %.1f us per pair"
          % (wl, a, b, ab, par), flush=True)
