"""In-kernel timeline of csrc/tail.hip (debug build only):
    make -C elektronn2_amd/csrc BUILD=build/dbg OUT=build/dbg/libe2hip.so DEBUG_ENV=1
    E2HIP_LIB=elektronn2_amd/csrc/build/dbg/libe2hip.so E2_TAIL_STAMPS=1 python tools/tail_stamps.py
prints the mean shader cycles a work-group spends between the kernel's stage boundaries, for
the tails of neuro3d_lite@183 and neuro3d@185, and the launch's duration (HIP events, 20 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from elektronn2_amd import backend

ctx = backend.Context(0)
for name, sp in (("lite183", (10, 37, 37)), ("full185", (5, 21, 21))):
    rng = np.random.RandomState(0)
    c1 = c2 = 200
    x = torch.tensor(rng.rand(1, c1, *sp).astype(np.float32), device="cuda")
    w1 = torch.tensor((rng.randn(c2, c1, 1, 1, 1) / np.sqrt(c1)).astype(np.float32), device="cuda")
    b1 = torch.tensor((rng.randn(c2) / 4).astype(np.float32), device="cuda")
    wh = torch.tensor((rng.randn(2, c2) / np.sqrt(c2)).astype(np.float32), device="cuda")
    bh = torch.zeros(2, device="cuda")
    t = torch.tensor(rng.randint(0, 2, (1, 1) + sp).astype(np.float32), device="cuda")
    wpf = torch.zeros(ctx.conv_ws_bytes(c2, c1, (1, 1, 1)) // 4 + 64, device="cuda")
    wpd = torch.zeros_like(wpf)
    ctx.conv3d_pack(w1, 0, wpf); ctx.conv3d_pack(w1, 1, wpd)
    probs = torch.empty((1, 2) + sp, device="cuda"); dpre = torch.empty((1, c2) + sp, device="cuda")
    dx = torch.empty((1, c1) + sp, device="cuda"); stats = torch.zeros(2, device="cuda")
    ws = torch.empty(ctx.tail_ws_bytes(x.shape, c2, 2) // 4 + 16, device="cuda")
    os.environ.pop("E2_TAIL_STAMPS", None)
    run = lambda: ctx.tail_fwd_bwd(x, wpf, wpd, b1, wh, bh, t, probs, dpre, dx, stats, ws)
    for _ in range(3):
        run()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(20):
        run()
    ctx.record(e1)
    torch.cuda.synchronize()
    print("%s: %.1f us per launch (back to back)" % (name, ctx.elapsed_ms(e0, e1) / 20 * 1e3), flush=True)
    os.environ["E2_TAIL_STAMPS"] = "1"
    run()
    torch.cuda.synchronize()
