"""In-kernel timeline of the position-split 1x1x1 weight gradient (csrc/conv_pw_wgrad.hip,
"MT,NT,8,0,S"; debug build only):
    make -C elektronn2_amd/csrc BUILD=build/dbg OUT=build/dbg/libe2hip.so DEBUG_ENV=1
    E2HIP_LIB=elektronn2_amd/csrc/build/dbg/libe2hip.so python tools/pwks_stamps.py [tiling ...]
prints, per tiling, the launch's duration (HIP events, accumulate: no fill) and the mean
s_memtime ticks a work-group spends between the kernel's stage boundaries, for the 200 -> 200
layers of neuro3d_lite@183 and neuro3d@185."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from elektronn2_amd import backend, autotune

ctx = backend.Context(0)
forced = [a for a in sys.argv[1:] if not a.startswith("sp=")]
shapes = [("lite183", (10, 37, 37)), ("full185", (5, 21, 21))]
for a in sys.argv[1:]:
    if a.startswith("sp="):                      # e.g. sp=8,40,43: rows of 13760 positions (64-byte aligned)
        shapes = [(a, tuple(int(v) for v in a[3:].split(",")))]
for name, sp in shapes:
    cin = cout = 200
    n = cout * sp[0] * sp[1] * sp[2]
    x = torch.rand(1, cin, *sp, device="cuda")
    flat = torch.zeros(n + 32, device="cuda")
    dy = flat[:n].view(1, cout, *sp)
    dy.copy_(torch.randn(1, cout, *sp, device="cuda"))
    dw = torch.zeros(cout, cin, 1, 1, 1, device="cuda")
    fn = lambda: ctx.conv3d_wgrad_pad(x, dy, dw, accumulate=True)
    cands = forced or [c for c in autotune.wgrad_candidates(cout, cin, (1, 1, 1), sp) if c.split(",")[2] == "8"]
    for c in cands:
        ctx.set_tiling("wgrad", c)
        os.environ.pop("E2_PWKS_STAMPS", None)
        try:
            t = autotune._time(ctx, fn, iters=20)
        except backend.E2Error as e:
            print(name, c, "refused:", e)
            continue
        torch.cuda.synchronize()
        print("%s %-16s %.1f us per launch (back to back, accumulate)" % (name, c, t * 1e3), flush=True)
        os.environ["E2_PWKS_STAMPS"] = "1"
        fn()
        torch.cuda.synchronize()
    ctx.set_tiling("wgrad", None)
