"""The captured training graph of a workload as a GraphViz file + its fork / join structure in text:
    python tools/graph_dot.py <workload> <f32|bf16> <out.dot>
(e2_graph_debug_dot; how DESIGN finding 54 looked at what stream capture made of the side stream)"""
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
import elektronn2_amd
from elektronn2_amd import nets, neuromancer as nm

wl, mode, out = sys.argv[1], sys.argv[2], sys.argv[3]
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
assert plan._graphs, "the step was not captured"
for i, g in enumerate(plan._graphs):
    path = out if i == 0 else out.replace(".dot", "_%d.dot" % i)
    plan.ctx.graph_debug_dot(g, path)
    txt = open(path).read()
    names = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
    edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
    print("graph %d: %d nodes, %d edges" % (i, len(names), len(edges)))
    def short(n):
        lab = names.get(n, n).replace("\\n", " ")
        return lab[:60]
    kids = {}
    for a, b in edges:
        kids.setdefault(a, []).append(b)
    for a, bs in kids.items():
        if len(bs) > 1:
            print("  fork %s -> %s" % (short(a), " | ".join(short(b) for b in bs)))
