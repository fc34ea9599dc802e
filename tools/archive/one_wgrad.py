"""30 launches of e2_conv3d_wgrad_bf16 on one layer with one tiling (for rocprofv3; see
tools/kstats_one_wgrad.sh):  python tools/one_wgrad.py cin cout kd kh kw D H W tile"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend

cin, cout, kd, kh, kw, D, H, W = map(int, sys.argv[1:9])
tile = sys.argv[9]
ctx = backend.Context(0)
k = (kd, kh, kw)
osp = (D - kd + 1, H - kh + 1, W - kw + 1)
x = torch.rand(1, cin, D, H, W, device="cuda")
dy = torch.randn(1, cout, *osp, device="cuda")
dw = torch.zeros(cout, cin, *k, device="cuda")
ctx.set_tiling("wgrad", tile)
for _ in range(30):
    ctx.conv3d_wgrad_bf16(x, dy, dw, accumulate=False)
torch.cuda.synchronize()
