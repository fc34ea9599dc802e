"""Which clock does the chip hold inside the conv GEMMs?  

usage (GPU box, debug-switch build of the library):
    E2HIP_LIB=elektronn2_amd/csrc/build/dbg/libe2hip.so python tools/clock_check.py [seconds]
and the same command under
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d <dir> -- python3 tools/clock_check.py

Per layer problem: `seconds` of back-to-back launches (the chip settles at the clock it
holds under this load; MI355X_MICROARCH.md "DVFS give-back" item 6), then ONE launch with
the in-kernel stamps switched on (E2_IGEMM_STAMPS, read per launch by the debug build):
every work-group stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start
and end, the library prints the median ratio = the in-kernel clock.  HIP events around the
back-to-back launches give us/launch; under rocprofv3 the counter file gives
GRBM_GUI_ACTIVE / 8 / duration for the same launches -- the "clock" tools/pmc_mfma.py
divides by."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elektronn2_amd import backend

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
# (op, cin, cout, k, input extent of the forward conv, tiling)   -- lite183 / full185 layers
CASES = [("fwd", 200, 200, (1, 3, 3), (10, 39, 39), "7,2,52,1"),
         ("dgrad", 40, 150, (2, 4, 4), (21, 44, 44), "3,4,8,3"),
         ("fwd", 40, 150, (2, 4, 4), (21, 44, 44), "2,4,16,1"),
         ("fwd", 40, 80, (4, 4, 4), (23, 39, 39), "5,2,40,1")]
ctx = backend.Context(0)
for op, cin, cout, k, sp, force in CASES:
    osp = tuple(sp[i] - k[i] + 1 for i in range(3))
    x = torch.rand(1, cin, *sp, device="cuda")
    w = torch.randn(cout, cin, *k, device="cuda") * 0.05
    y = torch.empty(1, cout, *osp, device="cuda")
    pshape = (1, cout) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
    dyp = torch.zeros(pshape, device="cuda")
    dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1],
        k[2] - 1:k[2] - 1 + osp[2]] = torch.randn(1, cout, *osp, device="cuda")
    dx = torch.empty_like(x)
    ws = torch.empty(ctx.conv_ws_bytes(cout, cin, k) // 4 + 64, device="cuda")
    ctx.conv3d_pack(w, 0 if op == "fwd" else 1, ws)
    fn = ((lambda: ctx.conv3d_fwd_packed(x, ws, cout, k, y)) if op == "fwd" else
          (lambda: ctx.conv3d_dgrad_packed(dyp, ws, cin, k, dx)))
    ctx.set_tiling("igemm", force)
    fn()
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    n, t0 = 0, time.time()
    ctx.record(e0)
    while time.time() - t0 < SECS:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()           # (keeps the launch queue short; ~1 % idle)
    ctx.record(e1)
    us = ctx.elapsed_ms(e0, e1) / n * 1e3
    gf = 2.0 * cout * cin * np.prod(k) * np.prod(osp) / 1e9
    print("%s %d->%d k%s in%s tiling %s: %.1f us/launch over %d back-to-back launches, "
          "%.1f TF = %.3f of 157.3" % (op, cin, cout, k, sp, force, us, n, gf / us * 1e3,
                                       gf / us * 1e3 / 157.3), flush=True)
    for _ in range(20):
        fn()
    os.environ["E2_IGEMM_STAMPS"] = "1"
    fn()                                   # the library prints the stamps + in-kernel clock
    del os.environ["E2_IGEMM_STAMPS"]
    sys.stderr.flush()
    ctx.set_tiling("igemm", None)
