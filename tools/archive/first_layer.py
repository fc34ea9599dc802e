"""The fused Cin = 1 first layer alone, repeatedly (for rocprofv3 PMC passes and A/B timing):
usage: python tools/first_layer.py <lite|full> [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend

which = sys.argv[1] if len(sys.argv) > 1 else "full"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
k, sp = ((1, 4, 4), (23, 183, 183)) if which == "lite" else ((1, 6, 6), (23, 185, 185))
ctx = backend.Context(0)
x = torch.rand(1, 1, *sp, device="cuda")
w = torch.randn(20, 1, *k, device="cuda") * 0.2
b = torch.rand(20, device="cuda") * 0.1
osp = (sp[0], (sp[1] - k[1] + 1) // 2, (sp[2] - k[2] + 1) // 2)
out = torch.empty(1, 20, *osp, device="cuda")
dout = torch.randn(1, 20, *osp, device="cuda")
dw, db = torch.zeros_like(w), torch.zeros(20, device="cuda")
fwd = lambda: ctx.conv1_pool_act_fwd(x, w, b, (1, 2, 2), 'relu', out)
bwd = lambda: ctx.conv1_pool_act_bwd(x, w, b, dout, (1, 2, 2), 'relu', dw, db)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(iters):
        fn()
    ctx.record(e1)
    print("%s first layer %s: %.1f us" % (which, name, ctx.elapsed_ms(e0, e1) / iters * 1e3))
