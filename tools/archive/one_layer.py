"""Run one conv layer op repeatedly (for rocprofv3 PMC / forced-config sweeps).
usage: python tools/one_layer.py <fwd|dgrad|wgrad> cin cout kd kh kw D H W [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend

op = sys.argv[1]
cin, cout, kd, kh, kw, D, H, W = map(int, sys.argv[2:10])
iters = int(sys.argv[10]) if len(sys.argv) > 10 else 10
ctx = backend.Context(0)
if os.environ.get('E2_MFMA_DTYPE'):
    ctx.set_mfma_dtype(os.environ['E2_MFMA_DTYPE'])
# tiling overrides of this tool (the library itself never reads the environment)
if os.environ.get("E2_IGEMM_FORCE"):
    ctx.set_tiling("igemm", os.environ["E2_IGEMM_FORCE"])
if os.environ.get("E2_WGRAD_FORCE"):
    ctx.set_tiling("wgrad", os.environ["E2_WGRAD_FORCE"])
k = (kd, kh, kw)
osp = (D - kd + 1, H - kh + 1, W - kw + 1)
x = torch.rand(1, cin, D, H, W, device="cuda")
w = torch.randn(cout, cin, *k, device="cuda") * 0.05
y = torch.empty(1, cout, *osp, device="cuda")
pshape = (1, cout) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
import numpy as np
flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
dyp = flat[:int(np.prod(pshape))].view(pshape)
dyp[:, :, kd - 1:kd - 1 + osp[0], kh - 1:kh - 1 + osp[1], kw - 1:kw - 1 + osp[2]] = torch.randn(1, cout, *osp, device="cuda")
dy = dyp[:, :, kd - 1:kd - 1 + osp[0], kh - 1:kh - 1 + osp[1], kw - 1:kw - 1 + osp[2]]
dx = torch.empty_like(x); dw = torch.empty_like(w)
ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
ctx.conv3d_pack(w, 0 if op == "fwd" else 1, ws)
fn = {"fwd": lambda: ctx.conv3d_fwd_packed(x, ws, cout, k, y),
      "dgrad": lambda: ctx.conv3d_dgrad_packed(dyp, ws, cin, k, dx),
      "wgrad": lambda: ctx.conv3d_wgrad(x, dy, dw),
      "wgradp": lambda: ctx.conv3d_wgrad_pad(x, dyp, dw)}[op]
for _ in range(3):
    fn()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(iters):
    fn()
ctx.record(e1)
us = ctx.elapsed_ms(e0, e1) / iters * 1e3
gf = 2.0 * cout * cin * kd * kh * kw * osp[0] * osp[1] * osp[2] / 1e9
print("%s %s: %.1f us  %.1f TF/s (%.1f%% of 157.3)  force=%s" % (
    op, sys.argv[2:10], us, gf / us * 1e3, gf / us * 1e3 / 157.3 * 100,
    os.environ.get("E2_IGEMM_FORCE") or os.environ.get("E2_WGRAD_FORCE")))
