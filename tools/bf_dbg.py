import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elektronn2_amd import backend
from oracle import e2_oracle as O
ctx = backend.Context(0)
def r(a): return torch.tensor(np.asarray(a, np.float32)).bfloat16().float().numpy()
np.set_printoptions(linewidth=200, precision=3, suppress=True)
for mode in ('f32', 'bf16'):
    ctx.set_mfma_dtype(mode)
    for (Ci, Co, k, sp) in [(16, 16, (1, 1, 1), (1, 1, 64))]:
        rng = np.random.RandomState(0)
        x = rng.rand(1, Ci, *sp).astype(np.float32)
        w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
        ref = O.conv3d_fwd(r(x), r(w)) if mode == 'bf16' else O.conv3d_fwd(x, w)
        ws = torch.empty(ctx.conv_ws_bytes(Co, Ci, k) // 4 + 64, device="cuda")
        ctx.conv3d_pack(torch.tensor(w, device="cuda"), 0, ws)
        y = torch.full(ref.shape, float("nan"), device="cuda")
        ctx.conv3d_fwd_packed(torch.tensor(x, device="cuda"), ws, Co, k, y)
        g = y.cpu().numpy()
        print(mode, "nan frac", np.isnan(g).mean())
        print((np.abs(g[0, :, 0, 0, :] - ref[0, :, 0, 0, :]) < 1e-4).astype(int))
        print(g[0, :4, 0, 0, :8]); print(ref[0, :4, 0, 0, :8])
