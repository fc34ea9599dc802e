"""Brute-force parity of every autotune candidate tiling on one small problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import backend, autotune
from oracle import e2_oracle as O

cin, cout, k, sp = int(sys.argv[1]), int(sys.argv[2]), tuple(map(int, sys.argv[3:6])), tuple(map(int, sys.argv[6:9]))
ctx = backend.Context(0)
rng = np.random.RandomState(0)
x = rng.rand(1, cin, *sp).astype(np.float32)
w = (rng.randn(cout, cin, *k) / 10).astype(np.float32)
y_ref = O.conv3d_fwd(x, w)
dy = rng.randn(*y_ref.shape).astype(np.float32)
dw_ref = O.conv3d_wgrad(dy, x, w.shape)
dx_ref = O.conv3d_dgrad(dy, w, x.shape)
xd, wd = torch.tensor(x).cuda(), torch.tensor(w).cuda()
pad = [kk - 1 for kk in k]
osp = y_ref.shape[2:]
dyp = torch.zeros(1, cout, *[osp[i] + 2 * pad[i] for i in range(3)], device="cuda")
inner = dyp[:, :, pad[0]:pad[0] + osp[0], pad[1]:pad[1] + osp[1], pad[2]:pad[2] + osp[2]]
inner.copy_(torch.tensor(dy).cuda())
def rel(a, b): return float(np.abs(a.cpu().numpy() - b).max() / np.abs(b).max())
bad = 0
for c in autotune.wgrad_candidates(cout, cin, k, osp):
    os.environ["E2_WGRAD_FORCE"] = c
    dw = torch.full(w.shape, float("nan"), device="cuda")
    try:
        ctx.conv3d_wgrad(xd, inner, dw)
    except backend.E2Error as e:
        continue
    r = rel(dw, dw_ref)
    if not r < 1e-4:
        bad += 1; print("WGRAD BAD", c, r)
os.environ.pop("E2_WGRAD_FORCE", None)
for c in autotune.igemm_candidates(cout, cin, k, osp):
    os.environ["E2_IGEMM_FORCE"] = c
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    try:
        ctx.conv3d_fwd(xd, wd, y)
    except backend.E2Error as e:
        continue
    r = rel(y, y_ref)
    if not r < 1e-4:
        bad += 1; print("FWD BAD", c, r)
for c in autotune.igemm_candidates(cin, cout, k, sp):
    os.environ["E2_IGEMM_FORCE"] = c
    dx = torch.full(x.shape, float("nan"), device="cuda")
    try:
        ctx.conv3d_dgrad(dyp, wd, dx)
    except backend.E2Error as e:
        continue
    r = rel(dx, dx_ref)
    if not r < 1e-4:
        bad += 1; print("DGRAD BAD", c, r)
print("done, bad =", bad)
