"""Parity of every HIP op (through the C ABI) against the CPU oracle.

Tolerance: BASELINE.json's north_star asks for 1e-4 relative in fp32; the fp32
MFMA path is an exact f32 fma chain, so these tests hold it to 2e-5 of the
reference's max magnitude (TOL) to catch indexing bugs that 1e-4 would hide.
"""
import os

import numpy as np
import pytest
import torch

from oracle import e2_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


def dev(a):
    return torch.tensor(np.asarray(a, np.float32), device="cuda")


def relerr(got, ref):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


CONV_CASES = [
    # (N, Cin, Cout, k, in spatial)
    (1, 1, 20, (1, 4, 4), (3, 21, 23)),
    (1, 20, 40, (3, 3, 3), (5, 14, 19)),
    (1, 40, 150, (2, 4, 4), (3, 12, 13)),
    (1, 150, 200, (1, 3, 3), (2, 11, 12)),
    (1, 200, 200, (1, 1, 1), (2, 9, 10)),
    (1, 200, 2, (1, 1, 1), (2, 9, 10)),
    (1, 1, 20, (1, 6, 6), (2, 25, 27)),
    (1, 20, 30, (1, 5, 5), (2, 17, 18)),
    (1, 40, 80, (4, 4, 4), (5, 9, 11)),
    (1, 80, 100, (3, 4, 4), (4, 8, 9)),
    (2, 3, 5, (2, 3, 2), (4, 7, 6)),
    (1, 6, 17, (1, 1, 3), (1, 1, 70)),      # single row, Q not /64
    (1, 30, 40, (1, 5, 5), (1, 47, 47)),     # Cin % 4 != 0, many row crossings
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "N%d_%d-%d_k%s_s%s" % (
    c[0], c[1], c[2], "x".join(map(str, c[3])), "x".join(map(str, c[4]))))
def test_conv3d_fwd_dgrad_wgrad(ctx, case):
    N, Ci, Co, k, sp = case
    rng = np.random.RandomState(hash(case) % (2 ** 31))
    x = rng.rand(N, Ci, *sp).astype(np.float32)
    w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    xd, wd = dev(x), dev(w)
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    ctx.conv3d_fwd(xd, wd, y)
    assert relerr(y, y_ref) < TOL

    # dgrad: dy lives in the interior of a zero-padded buffer
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    pad = [kk - 1 for kk in k]
    dyp = torch.zeros(N, Co, *[y_ref.shape[2 + i] + 2 * pad[i] for i in range(3)],
                      device="cuda")
    inner = dyp[:, :, pad[0]:pad[0] + y_ref.shape[2], pad[1]:pad[1] + y_ref.shape[3],
                pad[2]:pad[2] + y_ref.shape[4]]
    inner.copy_(dev(dy))
    dx = torch.full(x.shape, float("nan"), device="cuda")
    ctx.conv3d_dgrad(dyp, wd, dx)
    assert relerr(dx, O.conv3d_dgrad(dy, w, x.shape)) < TOL

    # wgrad reads the interior view (strided)
    dw = torch.full(w.shape, float("nan"), device="cuda")
    ctx.conv3d_wgrad(xd, inner, dw)
    assert relerr(dw, O.conv3d_wgrad(dy, x, w.shape)) < TOL


@pytest.mark.parametrize("force", ["1,1,4,1", "2,2,8,2", "3,1,4,4", "13,1,4,1", "10,2,8,2",
                                   "5,2,4,1", "7,1,8,1", "8,2,4,2", "4,1,8,1", "6,2,4,1",
                                   "7,2,24,1", "4,4,16,1", "13,1,24,3", "3,4,12,2"])
def test_conv3d_fwd_forced_tilings(ctx, force):
    """Every MT/NT instance, both CC values and split-K (atomic epilogue)."""
    rng = np.random.RandomState(7)
    x = rng.rand(1, 24, 5, 13, 37).astype(np.float32)
    w = (rng.randn(200, 24, 3, 2, 3) / 12).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_fwd(dev(x), dev(w), y)
    finally:
        ctx.set_tiling("igemm", None)
    assert relerr(y, y_ref) < TOL


def test_forced_tiling_that_the_launch_cannot_run_is_an_error(ctx):
    """e2hip.h, e2_set_tiling: "a tiling the problem cannot use makes the launch fail with
    an error, never silently fall back" -- also for "32,MB,NB" (the bf16-in-memory kernel's
    form) on a packed f32 launch, which used to fall through to the cost model"""
    from elektronn2_amd.backend import E2Error
    x = torch.rand(1, 8, 3, 9, 20, device="cuda")
    w = torch.rand(16, 8, 1, 3, 3, device="cuda")
    y = torch.empty(1, 16, 3, 7, 18, device="cuda")
    ctx.set_tiling("igemm", "32,2,2")
    try:
        with pytest.raises(E2Error, match="forced tiling"):
            ctx.conv3d_fwd(x, w, y)
    finally:
        ctx.set_tiling("igemm", None)
    ctx.conv3d_fwd(x, w, y)                   # the cost model's choice again
    assert relerr(y, O.conv3d_fwd(x.cpu().numpy(), w.cpu().numpy())) < TOL


def _padded_dy(dy, k):
    pad = [kk - 1 for kk in k]
    dyp = torch.zeros(dy.shape[0], dy.shape[1], *[dy.shape[2 + i] + 2 * pad[i] for i in range(3)],
                      device="cuda")
    dyp[:, :, pad[0]:pad[0] + dy.shape[2], pad[1]:pad[1] + dy.shape[3],
        pad[2]:pad[2] + dy.shape[4]].copy_(dev(dy))
    return dyp


@pytest.mark.parametrize("kw", [1, 3, 4, 5])
@pytest.mark.parametrize("force", ["5,4,8,1", "5,4,16,2", "5,4,24,3"])
def test_conv3d_5x4_block_tiling_fwd_and_dgrad(ctx, kw, force):
    """the widest 16x16x4 tile that ships (5 x 4 blocks: 80 accumulator registers, the
    instance the round-2 stream-K sweep faulted on) in its PLAIN form, every tap width,
    forward and data gradient, with and without split-K"""
    rng = np.random.RandomState(100 + kw)
    k = (2, 2, kw)
    x = rng.rand(1, 72, 4, 11, 40).astype(np.float32)
    w = (rng.randn(72, 72, *k) / np.sqrt(72 * np.prod(k))).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    dx = torch.full(x.shape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_fwd(dev(x), dev(w), y)
        ctx.conv3d_dgrad(_padded_dy(dy, k), dev(w), dx)
    finally:
        ctx.set_tiling("igemm", None)
    assert relerr(y, y_ref) < TOL
    assert relerr(dx, O.conv3d_dgrad(dy, w, x.shape)) < TOL


@pytest.mark.parametrize("force", [None, "2,1,8,1", "2,1,8,4", "3,2,8,6", "2,4,16,3", "1,2,4,8",
                                   "4,5,1,8,1,1,4,1", "4,5,1,8,4,1,4,1", "4,4,2,8,2,2,2,2"])
@pytest.mark.parametrize("k,do", [((4, 3, 3), 3), ((3, 2, 4), 1), ((2, 1, 5), 6)])
def test_conv3d_dgrad_skips_border_planes(ctx, force, k, do):
    """data gradient with kd > 1: the tap planes that read only the zero border of the
    padded gradient buffer are skipped per output plane (IgemmP::zpad) -- every dx plane
    still equals the oracle's, incl. split-K ranges that come out SHORTER or EMPTY on the
    border planes, one single real dy plane (do = 1), and the 4x4x1 kernel's item ring"""
    rng = np.random.RandomState(5 + do)
    ci, co = 20, 24
    dy = rng.randn(1, co, do, 7, 19).astype(np.float32)
    w = (rng.randn(co, ci, *k) / np.sqrt(ci * np.prod(k))).astype(np.float32)
    xshape = (1, ci, do + k[0] - 1, 7 + k[1] - 1, 19 + k[2] - 1)
    dx = torch.full(xshape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_dgrad(_padded_dy(dy, k), dev(w), dx)
    finally:
        ctx.set_tiling("igemm", None)
    assert relerr(dx, O.conv3d_dgrad(dy, w, xshape)) < TOL


@pytest.mark.parametrize("force", [None, "2,1,8,4", "3,2,8,6", "7,2,16,3", "2,4,16,3", "1,2,4,8",
                                   "2,2,8,1", "4,5,1,8,4,1,4,1", "4,4,2,8,2,2,2,2"])
@pytest.mark.parametrize("pool", [(1, 1, 1), (1, 2, 2), (2, 1, 1), (2, 2, 2)])
def test_split_k_partial_sums_are_added_up_by_the_consumer(ctx, force, pool):
    """split-K without atomics (e2hip.h): the conv stores one partial sum per K split into
    slabs that are NOT zeroed (filled with NaN here), the pooling / bias / activation
    kernel adds them up -- forward (the sum is left in part 0 for the backward) and
    backward (dout arrives as the partial sums of the consumer's data gradient, incl.
    border planes whose clipped K range leaves some splits EMPTY: they must store zeros).
    Every tiling form, split counts 1 .. 8, all four fixed pooling windows."""
    rng = np.random.RandomState(31)
    k = (3, 2, 3)
    ci, co = 22, 24
    osp = (4 * pool[0], 6 * pool[1], 8 * pool[2])
    x = rng.rand(1, ci, *[osp[i] + k[i] - 1 for i in range(3)]).astype(np.float32)
    w = (rng.randn(co, ci, *k) / np.sqrt(ci * np.prod(k))).astype(np.float32)
    b = (rng.randn(co) / 4).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    out_ref = O.bias_act_fwd(O.maxpool3d_fwd(y_ref, pool), b, 'relu')
    wd = dev(w)
    ws = torch.empty(ctx.conv_ws_bytes(co, ci, k) // 4 + 64, device="cuda")
    ctx.conv3d_pack(wd, 0, ws)
    yp = torch.full((8,) + y_ref.shape, float("nan"), device="cuda")
    out = torch.full(out_ref.shape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        n = ctx.conv3d_fwd_packed_parts(dev(x), ws, co, k, yp)
    finally:
        ctx.set_tiling("igemm", None)
    want = 1 if force is None else min(int(force.split(",")[4 if force.startswith("4,") else 3]), 8)
    if force is not None:      # (the 4x4x1 form re-balances the splits: <= the requested count)
        assert (n == want) if not force.startswith("4,") else (1 <= n <= want), (n, want)
    assert 1 <= n <= 8
    assert torch.isnan(yp[n:]).all()                          # untouched slabs
    ctx.pool_bias_act_fwd_parts(yp, n, dev(b), pool, 'relu', out)
    assert relerr(out, out_ref) < TOL
    assert relerr(yp[0], y_ref) < TOL                        # the sum, for the backward

    # the data gradient of THIS conv as partial sums -> the producer's activation backward
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref = O.conv3d_dgrad(dy, w, x.shape)                  # = dL/d(out of the producer)
    ws2 = torch.empty_like(ws)
    ctx.conv3d_pack(wd, 1, ws2)
    gp = torch.full((8,) + x.shape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        gn = ctx.conv3d_dgrad_packed_parts(_padded_dy(dy, k), ws2, ci, k, gp)
    finally:
        ctx.set_tiling("igemm", None)
    assert 1 <= gn <= 8 and torch.isnan(gp[gn:]).all() and not torch.isnan(gp[:gn]).any()
    assert relerr(gp[:gn].sum(0), dx_ref) < TOL
    # producer: a conv layer with this pooling window whose POOLED output is x
    ysh = (1, ci) + tuple(x.shape[2 + i] * pool[i] for i in range(3))
    yprod = rng.randn(*ysh).astype(np.float32)
    yprod[0, 0] = np.round(yprod[0, 0] * 2) / 2              # ties and exact zeros
    bp = (rng.randn(ci) / 4).astype(np.float32)
    bp[0] = 0.5
    pooled = O.maxpool3d_fwd(yprod, pool)
    dpre, db_ref = O.bias_act_bwd(dx_ref.astype(np.float32), pooled.astype(np.float32), bp, 'relu')
    dyp_ref = O.maxpool3d_bwd(dpre, yprod, pool)
    dyprod = torch.full(ysh, float("nan"), device="cuda")
    db = torch.zeros(ci, device="cuda")
    ctx.pool_bias_act_bwd_parts(gp, gn, dev(yprod), dev(bp), pool, 'relu', dyprod, db)
    assert relerr(dyprod, dyp_ref) < TOL
    assert relerr(db, db_ref) < 1e-4
    if pool == (1, 1, 1):       # ... or a fused-epilogue producer: slopes from its stored output
        outp = O.bias_act_fwd(yprod, bp, 'relu')
        outp_d = dev(outp)
        outp_d[dev(yprod + bp.reshape(1, -1, 1, 1, 1)) < 0] = -0.0
        d2 = torch.full(ysh, float("nan"), device="cuda")
        db2 = torch.zeros(ci, device="cuda")
        ctx.bias_act_bwd_out_parts(gp, gn, outp_d, 'relu', d2, db2)
        assert relerr(d2, dyp_ref) < TOL and relerr(db2, db_ref) < 1e-4


@pytest.mark.parametrize("act", ["relu", "lin"])
@pytest.mark.parametrize("k", [(1, 3, 3), (1, 1, 1), (2, 4, 4)])
def test_conv3d_fused_bias_act(ctx, act, k):
    """conv + bias + act in the GEMM epilogue (layers without pooling) and its
    backward from the activated output, incl. pre-activations that are EXACTLY zero
    (relu'(0) = 0.5) and negative (relu' = 0): channels 0 / 1 have zero weights."""
    rng = np.random.RandomState(21)
    x = rng.rand(2, 10, 4, 12, 22).astype(np.float32)
    w = (rng.randn(37, 10, *k) / 6).astype(np.float32)
    b = (rng.randn(37) / 4).astype(np.float32)
    w[0] = 0; b[0] = 0          # pre == 0 exactly
    w[1] = 0; b[1] = -1         # pre < 0 exactly
    out_ref, _ = O.conv_node_fwd(x, w, b, (1, 1, 1), act)
    wd = dev(w)
    ws = torch.empty(ctx.conv_ws_bytes(37, 10, k) // 4 + 64, device="cuda")
    ctx.conv3d_pack(wd, 0, ws)
    out = torch.full(out_ref.shape, float("nan"), device="cuda")
    ctx.conv3d_fwd_packed_act(dev(x), ws, 37, k, dev(b), act, out)
    assert relerr(out, out_ref) < TOL
    dout = rng.randn(*out_ref.shape).astype(np.float32)
    dy_ref, db_ref = O.bias_act_bwd(dout, O.conv3d_fwd(x, w), b, act)
    dy = torch.full(out_ref.shape, float("nan"), device="cuda")
    db = torch.zeros(37, device="cuda")
    ctx.bias_act_bwd_out(dev(dout), out, act, dy, db)
    assert relerr(dy, dy_ref) < TOL
    assert relerr(db, db_ref) < 1e-4
    if act == "relu":
        g = dy.cpu().numpy()
        assert np.array_equal(g[:, 0], 0.5 * dout[:, 0]) and not g[:, 1].any()


@pytest.mark.parametrize("force", ["4,4,2,8,1,4,3,1", "4,5,1,8,1,3,4,1", "4,5,2,16,1,1,4,2",
                                   "4,5,2,8,2,3,2,1", "4,7,1,8,1,2,6,1", "4,7,2,24,1,2,2,1",
                                   "4,8,1,16,1,2,4,1", "4,10,1,8,3,1,12,1", "4,13,1,8,1,1,4,2",
                                   "4,16,1,16,1,1,6,1", "4,5,1,8,1,6,2,1", "4,13,1,24,1,4,1,1"])
def test_conv3d_fwd_4x4_mfma_tilings(ctx, force):
    """the 4x4x1-MFMA kernel (igemm4_core.hpp): every (MG, NT) instance, WM x WN compute
    waves (4 .. 12 per work-group), one and two work-groups per CU walking the tiles,
    channel chunks with a ragged last chunk (Cin = 22), split-K, an odd Cout (50: not a
    multiple of 4), position tiles that straddle rows and planes' ends"""
    rng = np.random.RandomState(27)
    x = rng.rand(2, 22, 5, 13, 37).astype(np.float32)
    w = (rng.randn(50, 22, 3, 2, 3) / 12).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_fwd(dev(x), dev(w), y)
    finally:
        ctx.set_tiling("igemm", None)
    assert relerr(y, y_ref) < TOL


@pytest.mark.parametrize("kw,force", [(1, "4,13,1,16,1,2,4,1"), (1, "4,5,2,32,1,5,2,1"),
                                      (1, "4,7,2,16,2,4,1,1"), (4, "4,10,1,8,1,3,4,1"),
                                      (4, "4,16,1,12,2,2,2,2"), (5, "4,8,1,8,1,4,3,1"),
                                      (5, "4,5,1,4,1,5,2,1")])
def test_conv3d_4x4_mfma_tap_widths_dgrad_and_fused_act(ctx, kw, force):
    """tap rows of 1, 4 and 5 (3 is covered above): forward, data gradient on the padded
    gradient buffer (strided rows) and the fused bias + relu epilogue with its signed
    zeros, all through the 4x4x1 kernel"""
    rng = np.random.RandomState(kw)
    k = (1, 1, 1) if kw == 1 else (2, 3, kw)
    Ci, Co = 70, 100
    x = rng.rand(1, Ci, 4, 12, 21).astype(np.float32)
    w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
    b = rng.randn(Co).astype(np.float32) * 0.1
    y_ref = O.conv3d_fwd(x, w)
    ws = torch.empty(ctx.conv_ws_bytes(Co, Ci, k) // 4 + 64, device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_pack(dev(w), 0, ws)
        y = torch.full(y_ref.shape, float("nan"), device="cuda")
        ctx.conv3d_fwd_packed(dev(x), ws, Co, k, y)
        assert relerr(y, y_ref) < TOL
        if force.split(",")[4] == "1":                     # no split-K: fused epilogue
            ya = torch.full(y_ref.shape, float("nan"), device="cuda")
            ctx.conv3d_fwd_packed_act(dev(x), ws, Co, k, dev(b), 'relu', ya)
            pre = y_ref + b.reshape(1, -1, 1, 1, 1)
            assert relerr(ya, np.maximum(pre, 0)) < TOL
            neg = torch.signbit(ya).cpu().numpy()           # -0.0 marks a NEGATIVE pre-activation
            assert neg[pre < -1e-6].all() and not neg[pre > 1e-6].any()
        dy = rng.randn(*y_ref.shape).astype(np.float32)
        osp = y_ref.shape[2:]
        pshape = (1, Co) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
        flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
        dyp = flat[:int(np.prod(pshape))].view(pshape)
        dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1],
            k[2] - 1:k[2] - 1 + osp[2]] = dev(dy)
        ctx.conv3d_pack(dev(w), 1, ws)
        dx = torch.full(x.shape, float("nan"), device="cuda")
        # the data gradient has Cin = 70 output channels: 4*MG*WM may not cover it in one tile
        ctx.conv3d_dgrad_packed(dyp, ws, Ci, k, dx)
        assert relerr(dx, O.conv3d_dgrad(dy, w, x.shape)) < TOL
    finally:
        ctx.set_tiling("igemm", None)


@pytest.mark.parametrize("force", ["1,4,1", "1,4,2", "1,5,1", "1,5,2", "1,6,1", "1,6,2", "1,7,1", "1,7,2",
                                   "1,8,1", "1,8,2", "1,10,1", "1,10,2", "1,13,1", "1,13,2", "1,16,1",
                                   "1,4,1,64,0", "1,4,2,64,0", "1,7,1,128,0", "1,8,1,128,0", "1,13,2,64,0",
                                   "1,16,1,64,0", "1,5,1,128,0", "1,10,2,64,0"])
@pytest.mark.parametrize("Ci,Co,sp", [(70, 100, (4, 12, 21)), (200, 200, (2, 9, 37)), (33, 250, (1, 5, 70))])
def test_conv3d_pointwise_gemm(ctx, force, Ci, Co, sp):
    """csrc/conv_pw.hip ("1,MT,NT" / "1,MT,NT,KC,0"): the 1x1x1 conv as a GEMM with LDS-staged
    weights -- forward (batch of 2, input a strided interior view), the fused bias + relu
    epilogue with its signed zeros, the data gradient; channel counts that are not multiples
    of the 32 / 64 / 128-row chunks (a last chunk that runs past the packed image) or the
    16-row blocks, position counts that are not multiples of the tile"""
    rng = np.random.RandomState(Ci + Co)
    k = (1, 1, 1)
    N = 2
    x = rng.rand(N, Ci, *sp).astype(np.float32)
    w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci)).astype(np.float32)
    b = rng.randn(Co).astype(np.float32) * 0.1
    y_ref = O.conv3d_fwd(x, w)
    big = torch.zeros(N, Ci, sp[0] + 1, sp[1] + 2, sp[2] + 3, device="cuda")
    xv = big[:, :, 1:, 1:-1, 2:-1]
    xv.copy_(dev(x))
    ws = torch.empty(ctx.conv_ws_bytes(max(Co, Ci), max(Co, Ci), k) // 4 + 64, device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_pack(dev(w), 0, ws)
        y = torch.full(y_ref.shape, float("nan"), device="cuda")
        ctx.conv3d_fwd_packed(xv, ws, Co, k, y)
        assert relerr(y, y_ref) < TOL
        ya = torch.full(y_ref.shape, float("nan"), device="cuda")
        ctx.conv3d_fwd_packed_act(xv, ws, Co, k, dev(b), 'relu', ya)
        pre = y_ref + b.reshape(1, -1, 1, 1, 1)
        assert relerr(ya, np.maximum(pre, 0)) < TOL
        neg = torch.signbit(ya).cpu().numpy()               # -0.0 marks a NEGATIVE pre-activation
        assert neg[pre < -1e-6].all() and not neg[pre > 1e-6].any()
        dy = rng.randn(*y_ref.shape).astype(np.float32)
        ctx.conv3d_pack(dev(w), 1, ws)
        dx = torch.full(x.shape, float("nan"), device="cuda")
        ctx.conv3d_dgrad_packed(dev(dy), ws, Ci, k, dx)
        assert relerr(dx, O.conv3d_dgrad(dy, w, x.shape)) < TOL
    finally:
        ctx.set_tiling("igemm", None)


@pytest.mark.parametrize("force", [None, "2,2,8,1", "7,2,12,1", "3,4,8,1", "2,2,8,3", "5,1,4,2",
                                   "4,5,1,16,1,2,2,1"])
@pytest.mark.parametrize("k", [(1, 3, 3), (2, 4, 4), (1, 1, 1), (1, 2, 2)])
@pytest.mark.parametrize("mode", ["relu_out", "relu_pre", "lin"])
def test_conv3d_dgrad_fused_with_the_producers_activation_backward(ctx, force, k, mode):
    """e2_conv3d_dgrad_packed_actbwd: dgrad * act'(producer) straight into the interior of the
    producer's zero-padded gradient buffer + its bias gradient.  Wide epilogue (rows wider
    than Wo, pieces straddling row ends), split-K (atomics, whole-buffer zero fill), tilings
    without the epilogue (4x4x1 form, generic tap width: two launches, in place); slopes from
    the activated output (+0.0 -> 0.5, -0.0 -> 0) and from pre-activation + bias."""
    if force is not None and k[2] == 2:
        pytest.skip("generic tap width: the library's own tiling only")
    rng = np.random.RandomState(31)
    N, Cp, Co = 2, 23, 40                       # producer: Cp channels; this conv: Cp -> Co
    xsp = (4, 13, 22)
    w = (rng.randn(Co, Cp, *k) / 6).astype(np.float32)
    osp = tuple(xsp[i] - k[i] + 1 for i in range(3))
    dy = rng.randn(N, Co, *osp).astype(np.float32)
    pre = rng.randn(N, Cp, *xsp).astype(np.float32)          # producer pre-activation + bias
    bias = (rng.randn(Cp) / 4).astype(np.float32)
    pre[:, 0] = 0.0                                            # exactly zero: slope 0.5
    pre[:, 1] = -1.0
    dx_ref = O.conv3d_dgrad(dy, w, (N, Cp) + xsp)
    if mode == "lin":
        slope = np.ones_like(pre, dtype=np.float64)
    else:
        slope = np.where(pre > 0, 1.0, np.where(pre == 0, 0.5, 0.0))
    ref = dx_ref * slope
    db_ref = ref.sum(axis=(0, 2, 3, 4))
    # this conv's zero-padded gradient buffer
    pshape = (N, Co) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
    flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
    dyp = flat[:int(np.prod(pshape))].view(pshape)
    dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1],
        k[2] - 1:k[2] - 1 + osp[2]] = dev(dy)
    ws = torch.empty(ctx.conv_ws_bytes(Co, Cp, k) // 4 + 64, device="cuda")
    ctx.conv3d_pack(dev(w), 1, ws)
    # the producer's padded gradient buffer (its own kernel was (2,3,4): pads 1,2,3)
    pad = (1, 2, 3)
    qshape = (N, Cp) + tuple(xsp[i] + 2 * pad[i] for i in range(3))
    dq = torch.zeros(qshape, device="cuda")
    inner = dq[:, :, pad[0]:pad[0] + xsp[0], pad[1]:pad[1] + xsp[1], pad[2]:pad[2] + xsp[2]]
    inner.fill_(float("nan"))
    if mode == "relu_out":
        o = np.maximum(pre, 0)
        o[pre < 0] = -0.0
        src, act, bprev = dev(o.astype(np.float32)), "relu", None
    elif mode == "relu_pre":
        src, act, bprev = dev((pre - bias.reshape(1, -1, 1, 1, 1)).astype(np.float32)), "relu", dev(bias)
        # (pre - b) + b must reproduce the constructed zeros / signs exactly
        chk = (src.cpu().numpy() + bias.reshape(1, -1, 1, 1, 1)).astype(np.float32)
        slope = np.where(chk > 0, 1.0, np.where(chk == 0, 0.5, 0.0))
        ref = dx_ref * slope
        db_ref = ref.sum(axis=(0, 2, 3, 4))
    else:
        src, act, bprev = dev(pre), "lin", None
    db = torch.zeros(Cp, device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_dgrad_packed_actbwd(dyp, ws, Cp, k, src, act, dq, pad, db, bias_prev=bprev)
        assert relerr(inner, ref) < TOL
        assert relerr(db, db_ref) < 1e-4
        border = dq.clone()
        border[:, :, pad[0]:pad[0] + xsp[0], pad[1]:pad[1] + xsp[1], pad[2]:pad[2] + xsp[2]] = 0
        assert not border.any()                   # the padding stays zero
        if mode == "relu_out":
            assert relerr(inner[:, 0], 0.5 * dx_ref[:, 0]) < TOL and not inner[:, 1].any()
        # second call accumulates the bias gradient and rewrites the same interior
        ctx.conv3d_dgrad_packed_actbwd(dyp, ws, Cp, k, src, act, dq, pad, db, bias_prev=bprev)
        assert relerr(inner, ref) < TOL and relerr(db, 2 * db_ref) < 1e-4
        ctx.conv3d_dgrad_packed_actbwd(dyp, ws, Cp, k, src, act, dq, pad, None, bias_prev=bprev)
        assert relerr(inner, ref) < TOL and relerr(db, 2 * db_ref) < 1e-4
    finally:
        ctx.set_tiling("igemm", None)


@pytest.mark.parametrize("force", ["7,2,32,1", "13,1,16,2", "2,4,48,1", "5,2,64,1", "1,1,8,1"])
def test_conv3d_1x1_forced_tilings(ctx, force):
    """1x1x1 taps (plain GEMM): GU = 4 channel groups per step when CC % 16 == 0,
    ragged last chunk (Cin = 70)."""
    rng = np.random.RandomState(17)
    x = rng.rand(2, 70, 3, 11, 19).astype(np.float32)
    w = (rng.randn(100, 70, 1, 1, 1) / 8).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    ctx.set_tiling("igemm", force)
    try:
        ctx.conv3d_fwd(dev(x), dev(w), y)
    finally:
        ctx.set_tiling("igemm", None)
    assert relerr(y, y_ref) < TOL


@pytest.mark.parametrize("ci,co,k", [(200, 200, (1, 3, 3)), (150, 200, (1, 3, 3)), (40, 150, (2, 4, 4)),
                                     (100, 100, (3, 4, 4)), (80, 100, (3, 4, 4))])
def test_wgrad_pad_with_the_librarys_own_tiling(ctx, ci, co, k):
    """no tiling forced, nothing tuned: the cost model must only pick tilings that have an
    instance (it once chose 5..7 x 4 blocks, whose spilling instances had been removed)"""
    rng = np.random.RandomState(ci)
    x = rng.rand(1, ci, 3, 12, 14).astype(np.float32)
    osp = tuple(s - kk + 1 for s, kk in zip((3, 12, 14), k))
    dy = rng.randn(1, co, *osp).astype(np.float32)
    dw_ref = O.conv3d_wgrad(dy, x, (co, ci) + tuple(k))
    pad = [kk - 1 for kk in k]
    pshape = (1, co) + tuple(osp[i] + 2 * pad[i] for i in range(3))
    flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
    dyp = flat[:int(np.prod(pshape))].view(pshape)
    dyp[:, :, pad[0]:pad[0] + osp[0], pad[1]:pad[1] + osp[1], pad[2]:pad[2] + osp[2]] = dev(dy)
    dw = torch.full(dw_ref.shape, float("nan"), device="cuda")
    ctx.conv3d_wgrad_pad(dev(x), dyp, dw)
    assert relerr(dw, dw_ref) < TOL


def _plan_style_padded(dy, k):
    """the zero-padded gradient buffer as the launch plan lays it out (neuromancer/neural.py Conv
    _plan_alloc): rows OVERLAP -- the row pitch is the layer INPUT's width Wo + kw - 1"""
    N, C = dy.shape[:2]
    osp = dy.shape[2:]
    pad = [kk - 1 for kk in k]
    pshape = (N, C) + tuple(osp[i] + 2 * pad[i] for i in range(3))
    pitch = osp[2] + pad[2]
    plane = pshape[3] * pitch
    flat = torch.zeros(N * C * pshape[2] * plane + pad[2] + 32, device="cuda")
    flat[-32:] = float("nan")              # the slack may be read, never used
    dyp = flat.as_strided(pshape, (C * pshape[2] * plane, pshape[2] * plane, plane, pitch, 1))
    dyp[:, :, pad[0]:pad[0] + osp[0], pad[1]:pad[1] + osp[1], pad[2]:pad[2] + osp[2]] = dev(dy)
    return dyp


@pytest.mark.parametrize("force", ["13,2,9,0,1", "13,2,9,0,4", "7,2,9,0,3", "7,4,9,0,100000", "4,4,9,0,2",
                                   "10,2,9,0,5", "5,4,9,0,2", "3,4,9,0,7", "2,4,9,0,1", "8,2,9,0,3", "8,4,9,0,2",
                                   "6,4,9,0,4"])
@pytest.mark.parametrize("ci,co,k,sp", [(9, 100, (2, 3, 3), (4, 12, 21)), (40, 200, (1, 3, 3), (3, 11, 23)),
                                        (20, 37, (3, 3, 3), (5, 9, 18)), (33, 150, (2, 4, 4), (3, 14, 17)),
                                        (7, 30, (1, 5, 5), (2, 13, 16))])
def test_conv3d_wgrad_as_position_split_gemm(ctx, force, ci, co, k, sp):
    """csrc/conv_pw_wgrad.hip, "MT,NT,9,0,S": the weight gradient of a conv WITH taps as the
    K-contiguous GEMM of the 1x1x1 layers -- over the memory span of a gradient plane (dy at the
    input's row pitch, zeros in the gaps) every (input channel, tap) column of dW reads the
    channel's plane at a constant shift.  Units that run past the end of a plane (zeros of the
    border x anything readable), two samples, kd > 1, flipped tap order, channel counts off the
    tile sizes, one split to more splits than units, overwrite and accumulate; x as a channel slice
    of a wider tensor."""
    rng = np.random.RandomState(ci + co)
    N = 2
    x = rng.rand(N, ci, *sp).astype(np.float32)
    w = (rng.randn(co, ci, *k) / 8).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dw_ref = O.conv3d_wgrad(dy, x, w.shape)
    dyp = _plan_style_padded(dy, k)
    dw = torch.full(w.shape, float("nan"), device="cuda")
    # x followed by 32 finite floats (the caller's promise, e2_set_input_slack): huge ones here --
    # they meet the gradient's zero border only
    xflat = torch.full((x.size + 32,), 1e30, device="cuda")
    xd = xflat[:x.size].view(x.shape)
    xd.copy_(dev(x))
    ctx.set_tiling("wgrad", force)
    ctx.allowed_fallbacks = 1
    try:
        # without the promise the tiling is not offered: the library's own choice runs -- and
        # e2_last_launch says so (a fallback, another kernel family)
        ctx.conv3d_wgrad_pad(xd, dyp, dw)
        assert relerr(dw, dw_ref) < TOL
        fam, til, src = ctx.last_launch()
        assert src == "fallback" and fam != "wgrad_ks", (fam, til, src)
        ctx.set_input_slack(128)
        dw.fill_(float("nan"))
        ctx.conv3d_wgrad_pad(xd, dyp, dw)
        assert relerr(dw, dw_ref) < TOL
        assert ctx.last_launch() == ("wgrad_ks", force, "forced")     # pw_wgrad_ks_kernel<MT,NT,true> ran
        ctx.conv3d_wgrad_pad(xd, dyp, dw, accumulate=True)
        assert relerr(dw, 2 * dw_ref) < TOL
        assert ctx.last_launch() == ("wgrad_ks", force, "forced")
        wide = torch.full((N, ci + 4) + tuple(sp), 1e30, device="cuda")   # (a channel slice: more of the buffer follows)
        wide[:, 1:1 + ci] = dev(x)
        dw.fill_(float("nan"))
        ctx.conv3d_wgrad_pad(wide[:, 1:1 + ci], dyp, dw)
        assert relerr(dw, dw_ref) < TOL
        assert ctx.last_launch() == ("wgrad_ks", force, "forced")
    finally:
        ctx.set_input_slack(0)
        ctx.set_tiling("wgrad", None)


def test_an_unrunnable_forced_wgrad_tiling_is_reported_not_hidden(ctx):
    """"13,2,9,0,4" on a problem that does not qualify for the position-split
    GEMM -- (kh - 1) rows + kw - 1 = 11 zeros behind a gradient plane where 31 are needed -- takes
    the cost model's choice (the documented exemption of the forms 7 / 8 / 9, e2hip.h), and
    e2_last_launch / e2_tiling_fallbacks say so; any OTHER string the launch cannot run is an
    error, never a fallback"""
    from elektronn2_amd.backend import E2Error
    rng = np.random.RandomState(4)
    k = (1, 2, 2)
    x = rng.rand(1, 8, 2, 9, 10).astype(np.float32)
    dy = rng.randn(1, 16, 2, 8, 9).astype(np.float32)
    ref = O.conv3d_wgrad(dy, x, (16, 8) + k)
    dyp = _plan_style_padded(dy, k)
    xflat = torch.zeros(x.size + 32, device="cuda")
    xd = xflat[:x.size].view(x.shape)
    xd.copy_(dev(x))
    dw = torch.full(ref.shape, float("nan"), device="cuda")
    n0 = ctx.tiling_fallbacks()
    ctx.set_tiling("wgrad", "13,2,9,0,4")
    ctx.allowed_fallbacks = 1
    ctx.set_input_slack(128)
    try:
        ctx.conv3d_wgrad_pad(xd, dyp, dw)
        fam, til, src = ctx.last_launch()
        assert src == "fallback" and fam != "wgrad_ks" and til != "13,2,9,0,4", (fam, til, src)
        assert ctx.tiling_fallbacks() == n0 + 1
        assert relerr(dw, ref) < TOL
    finally:
        ctx.set_input_slack(0)
        ctx.set_tiling("wgrad", None)
    ctx.conv3d_wgrad_pad(xd, dyp, dw)                     # no string set: the cost model, by name
    assert ctx.last_launch()[2] == "model"
    ctx.set_tiling("wgrad", "6,3,1,128,2")                # no such instance of the direct kernel
    try:
        with pytest.raises(E2Error):
            ctx.conv3d_wgrad_pad(xd, dyp, dw)
    finally:
        ctx.set_tiling("wgrad", None)


@pytest.mark.parametrize("force", ["1,1,1,128,3", "2,2,1,256,5", "3,4,1,128,2", "4,1,1,256,7",
                                   "5,2,1,128,1", "7,2,1,128,3", "7,2,1,256,2", "7,1,1,128,40",
                                   "7,2,101,128,8", "2,2,101,256,2", "3,2,114,128,8", "1,1,101,128,8",
                                   "7,2,101,128,3", "2,2,101,256,5", "3,2,114,128,7",
                                   "3,4,14,256,3", "2,2,14,128,5", "7,2,14,256,1", "1,4,14,128,40"])
@pytest.mark.parametrize("k", [(2, 3, 3), (1, 1, 1), (1, 4, 1)])
def test_conv3d_wgrad_pad_forced_tilings(ctx, force, k):
    """direct kernel: dy from the zero-padded buffer; partial quads at the plane ends,
    N = 2, kd > 1, no padding at all (1x1x1), column padding only (1,4,1)."""
    rng = np.random.RandomState(9)
    x = rng.rand(2, 9, 4, 12, 21).astype(np.float32)
    w = (rng.randn(100, 9, *k) / 8).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dw_ref = O.conv3d_wgrad(dy, x, w.shape)
    osp = y_ref.shape[2:]
    pad = [kk - 1 for kk in k]
    pshape = (2, 100) + tuple(osp[i] + 2 * pad[i] for i in range(3))
    flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
    flat[-32:] = float("nan")              # the slack may be read, never used
    dev(dy)                                # (keeps the allocator from handing back zeros)
    dyp = flat[:int(np.prod(pshape))].view(pshape)
    dyp[:, :, pad[0]:pad[0] + osp[0], pad[1]:pad[1] + osp[1], pad[2]:pad[2] + osp[2]] = dev(dy)
    dw = torch.full(w.shape, float("nan"), device="cuda")
    ctx.set_tiling("wgrad", force)
    try:
        ctx.conv3d_wgrad_pad(dev(x), dyp, dw)
        assert relerr(dw, dw_ref) < TOL
        ctx.conv3d_wgrad_pad(dev(x), dyp, dw, accumulate=True)
        assert relerr(dw, 2 * dw_ref) < TOL
    finally:
        ctx.set_tiling("wgrad", None)


@pytest.mark.parametrize("force", ["2,2,7,0,1", "2,2,7,0,5", "4,2,7,0,3", "2,4,7,0,64", "4,4,7,0,2", "4,3,7,0,7",
                                   "3,4,7,0,1", "7,2,7,0,4", "2,7,7,0,9", "7,4,7,0,3", "4,7,7,0,100000",
                                   "13,2,8,0,1", "13,2,8,0,5", "13,2,8,0,100000", "7,2,8,0,3", "7,4,8,0,2",
                                   "4,4,8,0,7", "10,2,8,0,4"])
@pytest.mark.parametrize("Ci,Co,N,sp", [(70, 100, 2, (3, 7, 13)), (200, 200, 1, (2, 9, 37)), (33, 250, 1, (1, 5, 70)),
                                        (40, 37, 3, (1, 1, 5)), (200, 200, 2, (2, 8, 40)), (60, 208, 5, (1, 3, 23))])
def test_conv3d_pointwise_wgrad_gemm(ctx, force, Ci, Co, N, sp):
    """csrc/conv_pw_wgrad.hip ("MT,NT,7,0,S", and "MT,NT,8,0,S": the four waves of a work-group
    split the positions of ONE tile): the 1x1x1 weight gradient as a GEMM whose two
    operands are K-contiguous (positions): channel counts that are not multiples of the
    tiles, position counts that are not multiples of 16 / 32 (a masked last step per sample) or
    4, fewer than 32 positions, whole multiples of 32, batches (more samples than splits), one
    split (plain stores) to more splits than steps, overwrite and accumulate; channel rows that
    are NOT 16-byte aligned (odd plane sizes)."""
    rng = np.random.RandomState(Ci + Co)
    k = (1, 1, 1)
    x = rng.rand(N, Ci, *sp).astype(np.float32)
    dy = rng.randn(N, Co, *sp).astype(np.float32)
    ref = O.conv3d_wgrad(dy, x, (Co, Ci) + k)
    flat = torch.zeros(dy.size + 32, device="cuda")        # (the padded form of a 1x1x1 gradient
    dyp = flat[:dy.size].view(dy.shape)                   # is the gradient itself + its slack)
    dyp.copy_(dev(dy))
    dw = torch.full((Co, Ci) + k, float("nan"), device="cuda")
    fam = "pw_wgrad" if force.split(",")[2] == "7" else "pw_wgrad_ks"
    fb0 = ctx.tiling_fallbacks()
    ctx.set_tiling("wgrad", force)
    ctx.allowed_fallbacks = 2                 # (the two cropped views at the end)
    try:
        ctx.conv3d_wgrad_pad(dev(x), dyp, dw)
        assert relerr(dw, ref) < TOL
        assert ctx.last_launch() == (fam, force, "forced")
        ctx.conv3d_wgrad_pad(dev(x), dyp, dw, accumulate=True)
        assert relerr(dw, 2 * ref) < TOL
        # x as a channel slice of a wider buffer (a concat input): planes stay dense
        wide = torch.full((N, Ci + 5) + tuple(sp), float("nan"), device="cuda")
        wide[:, 3:3 + Ci] = dev(x)
        ctx.conv3d_wgrad_pad(wide[:, 3:3 + Ci], dyp, dw)
        assert relerr(dw, ref) < TOL
        assert ctx.last_launch() == (fam, force, "forced") and ctx.tiling_fallbacks() == fb0
        # views whose rows or planes are not dense (a crop of the parent: the tuning keys
        # hold the row pitch only, so a shipped "...,7,..." entry can meet one) are never
        # mis-read: the call takes the library's own tiling instead
        crop = torch.zeros((N, Ci, sp[0], sp[1], sp[2] + 2), device="cuda")
        crop[..., 1:-1] = dev(x)
        dw.fill_(float("nan"))
        ctx.conv3d_wgrad_pad(crop[..., 1:-1], dyp, dw)
        assert relerr(dw, ref) < TOL
        assert ctx.last_launch()[2] == "fallback" and not ctx.last_launch()[0].startswith("pw_wgrad")
        crop2 = torch.zeros((N, Ci, sp[0], sp[1] + 3, sp[2]), device="cuda")   # rows dense, planes not
        crop2[:, :, :, 2:-1] = dev(x)
        dw.fill_(float("nan"))
        ctx.conv3d_wgrad_pad(crop2[:, :, :, 2:-1], dyp, dw)
        assert relerr(dw, ref) < TOL
        assert ctx.last_launch()[2] == "fallback" and ctx.tiling_fallbacks() == fb0 + 2
    finally:
        ctx.set_tiling("wgrad", None)


@pytest.mark.parametrize("force", ["2,2,7,0,3", "4,4,7,0,1", "7,2,7,0,6", "13,2,8,0,3", "7,2,8,0,2", "4,4,8,0,1"])
def test_upconv3d_pointwise_wgrad_gemm(ctx, force):
    """UpConv's weight gradient through the same GEMM: rows m = co * R + r land in
    dw[co][ci][r]"""
    for case in UPCONV_CASES:
        ctx.set_tiling("wgrad", force)
        try:
            test_upconv3d(ctx, case)
        finally:
            ctx.set_tiling("wgrad", None)


@pytest.mark.parametrize("force", ["1,1,1,64,3", "2,2,1,128,5", "3,4,1,64,2", "4,1,4,128,7",
                                   "5,2,1,64,1", "7,2,1,128,3", "1,1,4,64,2", "7,1,1,64,4"])
def test_conv3d_wgrad_forced_tilings(ctx, force):
    rng = np.random.RandomState(8)
    x = rng.rand(2, 9, 4, 12, 21).astype(np.float32)
    dy = rng.randn(2, 100, 3, 11, 19).astype(np.float32)
    ref = O.conv3d_wgrad(dy, x, (100, 9, 2, 2, 3))
    dw = torch.full(ref.shape, float("nan"), device="cuda")
    ctx.set_tiling("wgrad", force)
    try:
        ctx.conv3d_wgrad(dev(x), dev(dy), dw)
    finally:
        ctx.set_tiling("wgrad", None)
    assert relerr(dw, ref) < TOL


def test_conv3d_strided_views(ctx):
    """x = crop of a bigger tensor, y = channel slice of a concat buffer."""
    rng = np.random.RandomState(3)
    big = rng.rand(1, 8, 7, 20, 22).astype(np.float32)
    w = (rng.randn(12, 8, 2, 3, 3) / 10).astype(np.float32)
    bd = dev(big)
    xv = bd[:, :, 1:6, 2:18, 3:20]
    y_ref = O.conv3d_fwd(big[:, :, 1:6, 2:18, 3:20], w)
    cat = torch.full((1, 20) + y_ref.shape[2:], float("nan"), device="cuda")
    ctx.conv3d_fwd(xv, dev(w), cat[:, 5:17])
    assert relerr(cat[:, 5:17], y_ref) < TOL
    assert torch.isnan(cat[:, :5]).all() and torch.isnan(cat[:, 17:]).all()


def test_conv3d_shape_errors(ctx):
    from elektronn2_amd.backend import E2Error
    x = torch.zeros(1, 2, 4, 8, 8, device="cuda")
    w = torch.zeros(3, 2, 2, 3, 3, device="cuda")
    with pytest.raises(E2Error):
        ctx.conv3d_fwd(x, w, torch.zeros(1, 3, 3, 6, 7, device="cuda"))
    with pytest.raises(TypeError):
        ctx.conv3d_fwd(x.double(), w, torch.zeros(1, 3, 3, 6, 6, device="cuda"))


POOL_CASES = [((1, 2, 2), 'relu'), ((2, 1, 1), 'relu'), ((2, 2, 2), 'relu'),
              ((1, 1, 1), 'relu'), ((1, 1, 1), 'lin'), ((1, 2, 2), 'lin'), ((3, 2, 1), 'relu')]


@pytest.mark.parametrize("pool,act", POOL_CASES)
def test_pool_bias_act(ctx, pool, act):
    rng = np.random.RandomState(11)
    shape = (2, 5, 6 * pool[0], 4 * pool[1], 5 * pool[2])
    y = rng.randn(*shape).astype(np.float32)
    # force ties inside windows and exact zeros of the pre-activation
    y[0, 0] = np.round(y[0, 0] * 2) / 2
    b = rng.randn(5).astype(np.float32)
    b[0] = 0.5
    p_ref = O.maxpool3d_fwd(y, pool)
    out_ref = O.bias_act_fwd(p_ref, b, act)
    out = torch.full(out_ref.shape, float("nan"), device="cuda")
    ctx.pool_bias_act_fwd(dev(y), dev(b), pool, act, out)
    assert relerr(out, out_ref) < 1e-7

    dout = rng.randn(*out_ref.shape).astype(np.float32)
    dp, db_ref = O.bias_act_bwd(dout, p_ref.astype(np.float32), b, act)
    dy_ref = O.maxpool3d_bwd(dp, y, pool)
    dy = torch.full(y.shape, float("nan"), device="cuda")
    db = torch.zeros(5, device="cuda")
    ctx.pool_bias_act_bwd(dev(dout), dev(y), dev(b), pool, act, dy, db)
    assert relerr(dy, dy_ref) < 1e-7
    assert relerr(db, db_ref) < 1e-5


def test_maxpool_standalone_and_floor(ctx):
    rng = np.random.RandomState(12)
    x = rng.randn(1, 3, 5, 7, 9).astype(np.float32)     # not divisible: floor semantics
    pool = (2, 2, 2)
    ref = O.maxpool3d_fwd(x, pool)
    out = torch.empty(ref.shape, device="cuda")
    ctx.maxpool3d_fwd(dev(x), pool, out)
    assert relerr(out, ref) == 0.0
    dout = rng.randn(*ref.shape).astype(np.float32)
    dx = torch.full(x.shape, float("nan"), device="cuda")
    ctx.maxpool3d_bwd(dev(dout), dev(x), pool, dx)
    assert relerr(dx, O.maxpool3d_bwd(dout, x, pool)) < 1e-7
    base = rng.randn(*x.shape).astype(np.float32)
    dx2 = dev(base)
    ctx.maxpool3d_bwd(dev(dout), dev(x), pool, dx2, accumulate=True)
    assert relerr(dx2, base + O.maxpool3d_bwd(dout, x, pool)) < 1e-6


UPCONV_CASES = [(1, 8, 6, (2, 2, 2), (3, 4, 5), 'relu'), (1, 32, 16, (1, 2, 2), (2, 5, 6), 'relu'),
                (2, 5, 7, (2, 1, 3), (2, 3, 4), 'lin'), (1, 64, 64, (2, 2, 2), (2, 9, 9), 'relu')]


@pytest.mark.parametrize("force", ["4,16,1,16,1,2,2,1", "4,13,1,16,1,4,3,1", "4,4,2,16,2,2,3,2"])
def test_upconv3d_4x4_mfma_scatter_epilogue(ctx, force):
    """UpConv forward through the 4x4x1 kernel: 1x1x1 GEMM to Cout * prod(pool) rows with the
    depth-to-space scatter in the epilogue (the backward's data gradient takes the same kernel)"""
    for case in UPCONV_CASES:
        ctx.set_tiling("igemm", force)
        try:
            test_upconv3d(ctx, case)
        finally:
            ctx.set_tiling("igemm", None)


@pytest.mark.parametrize("force", ["1,4,1", "1,4,2", "1,8,1,64,0", "1,7,2,64,0", "1,5,1,128,0", "1,6,2"])
def test_upconv3d_pointwise_gemm_scatter_epilogue(ctx, force):
    """UpConv forward through csrc/conv_pw.hip: M = Cout * prod(pool) rows, the depth-to-space
    scatter and bias[row / R] + activation in the GEMM's epilogue (no separate bias launch);
    the backward's data gradient takes the same kernel in its plain form"""
    for case in UPCONV_CASES:
        ctx.set_tiling("igemm", force)
        try:
            test_upconv3d(ctx, case)
        finally:
            ctx.set_tiling("igemm", None)


@pytest.mark.parametrize("case", UPCONV_CASES)
def test_upconv3d(ctx, case):
    N, Ci, Co, pool, sp, act = case
    rng = np.random.RandomState(13)
    x = rng.randn(N, Ci, *sp).astype(np.float32)
    w = (rng.randn(Co, Ci, *pool) / np.sqrt(Ci)).astype(np.float32)
    b = rng.randn(Co).astype(np.float32)
    y_ref = O.bias_act_fwd(O.upconv3d_fwd(x, w, pool), b, act)
    assert np.abs(O.upconv3d_fwd(x, w, pool) - O.upconv3d_fwd_literal(x, w, pool)).max() < 1e-12
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    ctx.upconv3d_fwd(dev(x), dev(w), dev(b), pool, act, y)
    assert relerr(y, y_ref) < TOL
    dout = rng.randn(*y_ref.shape).astype(np.float32)
    dpre, db_ref = O.bias_act_bwd(dout, O.upconv3d_fwd(x, w, pool), b, act)
    dx = torch.full(x.shape, float("nan"), device="cuda")
    dw = torch.full(w.shape, float("nan"), device="cuda")
    db = torch.full((Co,), float("nan"), device="cuda")
    ctx.upconv3d_bwd(dev(x), dev(w), y, dev(dout), pool, act, dx, dw, db)
    assert relerr(dx, O.upconv3d_dgrad(dpre, w, pool)) < TOL
    assert relerr(dw, O.upconv3d_wgrad(dpre, x, pool)) < TOL
    assert relerr(db, db_ref) < TOL
    # the form the training plan uses: both images kept current by ONE repack launch for all
    # layers (pack jobs of mode 2 / 3 next to a conv's), the output a channel slice of a
    # wider buffer (the concat it feeds), dw / dbias ADDED to a zeroed gradient arena
    nb = ctx.upconv_image_bytes(Co, Ci, pool) // 4 + 64
    wp_f = torch.zeros(nb, device="cuda")
    wp_d = torch.zeros(nb, device="cuda")
    wc = (rng.randn(7, 5, 1, 3, 3)).astype(np.float32)            # a conv job in the same launch
    wpc = torch.zeros(ctx.conv_ws_bytes(7, 5, (1, 3, 3)) // 4 + 64, device="cuda")
    wd, wcd = dev(w), dev(wc)
    jobs = ctx.make_pack_jobs([(wd, wp_f, 2), (wcd, wpc, 0), (wd, wp_d, 3)])
    ctx.conv3d_pack_multi(*jobs)
    ref_img = torch.zeros_like(wpc)
    ctx.conv3d_pack(wcd, 0, ref_img)
    assert torch.equal(wpc, ref_img)                    # the conv job is untouched by its neighbours
    wide = torch.full((N, Co + 3) + tuple(y_ref.shape[2:]), float("nan"), device="cuda")
    yv = wide[:, 2:2 + Co]
    ctx.upconv3d_fwd_packed(dev(x), wp_f, dev(b), Co, pool, act, yv)
    assert torch.equal(yv, y)                           # same GEMM, same image: bit for bit
    gw = torch.full((N, Co + 3) + tuple(y_ref.shape[2:]), float("nan"), device="cuda")
    gv = gw[:, 2:2 + Co]
    gv.copy_(dev(dout))
    ws = torch.empty(ctx.upconv_ws_bytes(Co, Ci, pool, x.shape) // 4 + 64, device="cuda")
    dx2 = torch.full(x.shape, float("nan"), device="cuda")
    dw2 = torch.full(w.shape, 1.0, device="cuda")
    db2 = torch.full((Co,), 2.0, device="cuda")
    ctx.upconv3d_bwd_packed(dev(x), wp_d, yv, gv, pool, act, dx2, dw2, db2, ws, accumulate=True)
    assert relerr(dx2, O.upconv3d_dgrad(dpre, w, pool)) < TOL
    assert relerr(dw2, O.upconv3d_wgrad(dpre, x, pool) + 1.0) < TOL
    assert relerr(db2, db_ref + 2.0) < TOL
    ctx.upconv3d_bwd_packed(dev(x), wp_d, yv, gv, pool, act, dx2, dw2, db2, ws)   # overwrite form
    assert relerr(dw2, O.upconv3d_wgrad(dpre, x, pool)) < TOL
    assert relerr(db2, db_ref) < TOL
    # an UpConv whose parent needs no gradient (directly on an Input): no dx, no image --
    # the weight-gradient half reads neither (this call was refused)
    dw3 = torch.full(w.shape, float("nan"), device="cuda")
    db3 = torch.full((Co,), float("nan"), device="cuda")
    ctx.upconv3d_bwd_packed(dev(x), None, yv, gv, pool, act, None, dw3, db3, ws)
    assert relerr(dw3, O.upconv3d_wgrad(dpre, x, pool)) < TOL
    assert relerr(db3, db_ref) < TOL


def test_transposes_and_copy(ctx):
    rng = np.random.RandomState(14)
    big = rng.randn(2, 37, 5, 9, 13).astype(np.float32)
    bd = dev(big)
    v = bd[:, 3:36, 1:4, 2:9, 1:12]
    ref = big[:, 3:36, 1:4, 2:9, 1:12]
    nd = torch.empty(ref.transpose(0, 2, 3, 4, 1).shape, device="cuda")
    ctx.to_ndhwc(v, nd)
    assert relerr(nd, ref.transpose(0, 2, 3, 4, 1)) == 0.0
    back = torch.full(ref.shape, float("nan"), device="cuda")
    ctx.to_ncdhw(nd, back)
    assert relerr(back, ref) == 0.0
    dst = torch.ones(ref.shape, device="cuda")
    ctx.copy5(v, dst, accumulate=True)
    assert relerr(dst, ref + 1) == 0.0
    f = torch.empty(1000, device="cuda")
    ctx.fill(f, 2.5)
    assert float(f.min()) == 2.5 and float(f.max()) == 2.5


def test_softmax_nll(ctx):
    rng = np.random.RandomState(15)
    lg = (rng.randn(2, 3, 4, 5, 6) * 3).astype(np.float32)
    tg = rng.randint(0, 3, (2, 1, 4, 5, 6)).astype(np.float32)
    tg[0, 0, 0, 0, :3] = -1            # unlabelled voxels are ignored
    loss_ref, dl_ref, p_ref = O.nll_loss_and_grad(lg, tg)
    probs = torch.empty(lg.shape, device="cuda")
    stats = torch.zeros(2, device="cuda")
    ctx.softmax_nll_fwd(dev(lg), dev(tg), probs, stats)
    assert relerr(probs, p_ref) < 1e-6
    dl = torch.empty(lg.shape, device="cuda")
    loss = torch.zeros(1, device="cuda")
    ctx.softmax_nll_bwd(probs, dev(tg), stats, dl, loss)
    assert abs(float(loss) - loss_ref) / loss_ref < 1e-5
    assert relerr(dl, dl_ref) < 1e-5
    assert abs(float(stats[1]) - (tg >= 0).sum()) < 0.5


@pytest.mark.parametrize("n", [5000, 300003])
def test_adam_and_sgd(ctx, n):
    """(float4 body + scalar tail, several work-groups: the step counter t is published by
    the last work-group to finish and must read 3 after three steps)"""
    rng = np.random.RandomState(16)
    p0 = rng.randn(n); g = rng.randn(n)
    seg = np.array([0, 1200, 3000, n], np.int64)
    reg = np.array([1.0, 0.0, 3.0], np.float32)
    lr, mom, b2, wd = 5e-4, 0.9, 0.999, 0.5e-4
    p = dev(p0); m = torch.zeros(n, device="cuda"); s = torch.zeros(n, device="cuda")
    hyper = dev([lr, mom, b2, wd, 0, 0, 0, 0])
    so = torch.tensor(seg, device="cuda"); sr = dev(reg)
    pr, mr, sr_ = p0.astype(np.float32).astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 4):
        ctx.adam_step(p, dev(g * t), m, s, so, sr, hyper)
        out = np.empty(n)
        for k in range(3):
            sl = slice(seg[k], seg[k + 1])
            out[sl], mr[sl], sr_[sl] = O.adam_step(pr[sl], (g * t)[sl].astype(np.float32), mr[sl],
                                                   sr_[sl], t, lr, mom, b2, wd, reg[k])
        pr = out
    assert relerr(p, pr) < 1e-5
    assert float(hyper[4]) == 3.0
    d = torch.zeros(n, device="cuda"); p = dev(p0)
    ctx.sgd_step(p, dev(g), d, so, sr, hyper)
    out = np.empty(n)
    for k in range(3):
        sl = slice(seg[k], seg[k + 1])
        out[sl], _ = O.sgd_step(p0.astype(np.float32)[sl], g.astype(np.float32)[sl], 0.0, lr, mom, wd, reg[k])
    assert relerr(p, out) < 1e-6


@pytest.mark.parametrize("gdiv", [False, True])
def test_adam_step_that_writes_the_packed_images(ctx, gdiv):
    """e2_adam_pack_step (csrc/update_pack.hip): the Adam update of optimiser.py:301-329 and the
    repack of every conv's weight images in ONE launch.  Against the two launches it replaces --
    e2_adam_step_ex, then e2_conv3d_pack_multi on the updated weights -- parameters, both moments,
    the cleared gradient arena, the published step counter AND both images of every tensor are
    bit-identical: kernels with 1, 9, 16, 25 taps per plane and 1-3 planes, channel counts off
    the 32-channel tiles, a tensor with a forward image only, tensors without images (biases, a
    first-layer weight) between them, weight decay on and off, the data-parallel scaling
    (gmul / gdiv)."""
    rng = np.random.RandomState(21)
    convs = [(40, 20, 3, 3, 3, True, True), (200, 150, 1, 3, 3, True, True), (100, 80, 3, 4, 4, True, True),
             (40, 30, 1, 5, 5, True, False), (200, 200, 1, 1, 1, True, True), (37, 21, 2, 2, 3, False, True)]
    # arena: conv weight, its bias, ... + one tensor without images at the front (a first layer)
    tensors = [("w0", (20, 1, 1, 4, 4), 1.0, None)]
    for i, (co, ci, kd, kh, kw, f, d) in enumerate(convs):
        tensors.append(("w%d" % (i + 1), (co, ci, kd, kh, kw), 1.0 if i != 2 else 0.0, (f, d)))
        tensors.append(("b%d" % (i + 1), (co,), 0.0, None))
    offs, off = {}, 0
    for name, sh, reg, _ in tensors:
        offs[name] = off
        off += (int(np.prod(sh)) + 3) // 4 * 4
    n = off
    P0 = np.zeros(n, np.float32); G0 = np.zeros(n, np.float32)
    M0 = np.zeros(n, np.float32); S0 = np.zeros(n, np.float32)
    seg_off, seg_reg = [], []
    for name, sh, reg, _ in tensors:
        o, c = offs[name], int(np.prod(sh))
        P0[o:o + c] = rng.randn(c) * 0.1; G0[o:o + c] = rng.randn(c) * 0.01
        M0[o:o + c] = rng.randn(c) * 0.01; S0[o:o + c] = rng.rand(c) * 1e-3
        seg_off.append(o); seg_reg.append(reg)
    seg_off.append(n)
    hyp = [5e-4, 0.9, 0.999, 0.5e-4, 3, 0, 0, 0] + [0] * 16     # (e2_adam_pack_step: 24 floats)
    count = dev([1234.0]) if gdiv else None
    kw = dict(gdiv=count, gmul=1.0 if gdiv else 0.125, zero_g=True)

    def images(Pdev):
        """zero-filled images + the pack jobs / update jobs that refer to them"""
        pj, uj, imgs = [], [], {}
        for name, sh, reg, im in tensors:
            if im is None:
                continue
            co, ci, kd, kh, kwd = sh
            w = Pdev[offs[name]:offs[name] + int(np.prod(sh))].view(sh)
            nb = ctx.conv_ws_bytes(co, ci, (kd, kh, kwd))
            wf = torch.zeros(nb // 4 + 64, device="cuda") if im[0] else None
            wd_ = torch.zeros(nb // 4 + 64, device="cuda") if im[1] else None
            if wf is not None:
                pj.append((w, wf, 0))
            if wd_ is not None:
                pj.append((w, wd_, 1))
            uj.append((offs[name], wf, wd_, sh, reg))
            imgs[name] = (wf, wd_)
        return pj, uj, imgs

    # the two launches
    Pr, Gr, Mr, Sr, Hr = dev(P0), dev(G0), dev(M0), dev(S0), dev(hyp)
    pj, _, img_ref = images(Pr)
    ctx.adam_step(Pr, Gr, Mr, Sr, torch.tensor(seg_off, device="cuda"), dev(seg_reg), Hr, **kw)
    ctx.conv3d_pack_multi(*ctx.make_pack_jobs(pj))
    # the one launch
    Pf, Gf, Mf, Sf, Hf = dev(P0), dev(G0), dev(M0), dev(S0), dev(hyp)
    _, uj, img = images(Pf)
    rests = [(offs[name], int(np.prod(sh)), reg) for name, sh, reg, im in tensors if im is None]
    upd = ctx.make_upd_jobs(uj, rests)
    ctx.adam_pack_step(Pf, Gf, Mf, Sf, upd, Hf, **kw)
    torch.cuda.synchronize()
    same = lambda a, b: torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert same(Pf, Pr) and same(Mf, Mr) and same(Sf, Sr)
    assert float(Gf.abs().max()) == 0.0 and float(Gr.abs().max()) == 0.0
    assert float(Hf[4]) == 4.0 and same(Hf[:6], Hr[:6])
    assert not torch.equal(Pf, dev(P0))
    for name, (wf, wd_) in img.items():
        rf, rd = img_ref[name]
        if wf is not None:
            assert float(wf.abs().max()) > 0 and same(wf, rf), name
        if wd_ is not None:
            assert float(wd_.abs().max()) > 0 and same(wd_, rd), name
    # a second step on top (t = 5): still the same as the two launches
    Gr.copy_(dev(G0)); Gf.copy_(dev(G0))
    ctx.adam_step(Pr, Gr, Mr, Sr, torch.tensor(seg_off, device="cuda"), dev(seg_reg), Hr, **kw)
    ctx.conv3d_pack_multi(*ctx.make_pack_jobs(pj))
    ctx.adam_pack_step(Pf, Gf, Mf, Sf, upd, Hf, **kw)
    assert same(Pf, Pr) and float(Hf[4]) == 5.0
    for name, (wf, wd_) in img.items():
        assert (wf is None or same(wf, img_ref[name][0])) and (wd_ is None or same(wd_, img_ref[name][1])), name


def test_graph_capture_replay(ctx):
    """e2_graph_*: capture conv+pool on a side stream, replay twice."""
    rng = np.random.RandomState(17)
    x = rng.rand(1, 4, 3, 10, 12).astype(np.float32)
    w = rng.randn(6, 4, 1, 3, 3).astype(np.float32)
    b = rng.randn(6).astype(np.float32)
    ref, _ = O.conv_node_fwd(x, w, b, (1, 2, 2), 'relu')
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty(1, 6, 3, 8, 10, device="cuda")
    out = torch.zeros(ref.shape, device="cuda")
    ws = torch.empty(ctx.conv_ws_bytes(6, 4, (1, 3, 3)) // 4 + 64, device="cuda")
    side = torch.cuda.Stream()
    old = ctx.stream
    torch.cuda.synchronize()
    ctx.set_stream(side)
    try:
        ctx.graph_begin()
        ctx.conv3d_fwd(xd, wd, y, ws)
        ctx.pool_bias_act_fwd(y, bd, (1, 2, 2), 'relu', out)
        g = ctx.graph_end()
        for _ in range(2):
            out.zero_()
            torch.cuda.synchronize()
            ctx.graph_launch(g)
            ctx.synchronize()
            assert relerr(out, ref) < TOL
        ctx.graph_destroy(g)
    finally:
        ctx.set_stream(old)


@pytest.mark.parametrize("k,sp,pool", [((1, 4, 4), (3, 23, 71), (1, 2, 2)),
                                       ((1, 6, 6), (2, 25, 77), (1, 2, 2)),
                                       ((1, 4, 4), (1, 19, 11), (1, 2, 2)),
                                       ((1, 3, 3), (3, 21, 70), (1, 1, 1))])
def test_fused_first_layer(ctx, k, sp, pool):
    """csrc/conv_first.hip: conv(1 ch) -> pool(1,2,2) or none -> +b -> relu, fwd and bwd
    (recompute)."""
    rng = np.random.RandomState(21)
    x = rng.rand(2, 1, *sp).astype(np.float32)
    w = (rng.randn(20, 1, *k) / 3).astype(np.float32)
    b = (rng.randn(20) / 4).astype(np.float32)
    assert ctx.conv1_supported(1, k, pool) and not ctx.conv1_supported(2, k, pool)
    out_ref, cache = O.conv_node_fwd(x, w, b, pool, 'relu')
    out = torch.full(out_ref.shape, float("nan"), device="cuda")
    ctx.conv1_pool_act_fwd(dev(x), dev(w), dev(b), pool, 'relu', out)
    assert relerr(out, out_ref) < TOL
    dout = rng.randn(*out_ref.shape).astype(np.float32)
    _, dw_ref, db_ref = O.conv_node_bwd(dout, x, w, b, cache, pool, 'relu', need_dx=False)
    dw = torch.zeros(w.shape, device="cuda"); db = torch.zeros(20, device="cuda")
    ctx.conv1_pool_act_bwd(dev(x), dev(w), dev(b), dev(dout), pool, 'relu', dw, db)
    assert relerr(dw, dw_ref) < TOL
    assert relerr(db, db_ref) < TOL


@pytest.mark.parametrize("ncls,cin", [(2, 200), (3, 37), (4, 300)])
def test_fused_classifier_head(ctx, ncls, cin):
    """1x1x1 'lin' conv -> softmax -> MultinoulliNLL fused (csrc/head.hip): probs, loss
    and all gradients vs the oracle; unlabelled voxels (target -1), strided input view,
    a position count that is not a multiple of the 64-position tile, dx accumulation."""
    rng = np.random.RandomState(31)
    sp = (3, 7, 13)
    xfull = rng.rand(2, cin + 3, *sp).astype(np.float32)
    x = xfull[:, 2:2 + cin]                            # channel-sliced view
    w = (rng.randn(ncls, cin, 1, 1, 1) / np.sqrt(cin)).astype(np.float32)
    b = (rng.randn(ncls) / 4).astype(np.float32)
    t = rng.randint(-1, ncls, (2, 1) + sp).astype(np.float32)
    logits = O.conv3d_fwd(x, w) + b.reshape(1, -1, 1, 1, 1)
    loss_ref, dlog, p_ref = O.nll_loss_and_grad(logits, t)
    dw_ref = O.conv3d_wgrad(dlog, x, w.shape)
    db_ref = dlog.sum(axis=(0, 2, 3, 4))
    dx_ref = O.conv3d_dgrad(dlog, w, x.shape)
    xd = dev(xfull)[:, 2:2 + cin]
    probs = torch.full((2, ncls) + sp, float("nan"), device="cuda")
    stats = torch.zeros(2, device="cuda")
    ctx.head_fwd(xd, dev(w), dev(b), dev(t), probs, stats)
    assert relerr(probs, p_ref) < TOL
    probs2 = torch.full((2, ncls) + sp, float("nan"), device="cuda")
    ctx.head_fwd(xd, dev(w), dev(b), None, probs2, None)      # prediction only
    assert torch.equal(probs, probs2)
    dx = torch.full(x.shape, 1e-3, device="cuda")
    dw = torch.zeros(w.shape, device="cuda")
    db = torch.zeros(ncls, device="cuda")
    loss = torch.zeros(1, device="cuda")
    ctx.head_bwd(xd, dev(w), probs, dev(t), stats, dx, True, dw, db, loss)
    assert abs(float(loss) - loss_ref) / loss_ref < 1e-5
    assert relerr(dx - 1e-3, dx_ref) < 1e-3              # accumulated onto a constant
    assert relerr(dw, dw_ref) < TOL
    assert relerr(db, db_ref) < 1e-4
    dx2 = torch.full(x.shape, float("nan"), device="cuda")
    ctx.head_bwd(xd, dev(w), probs, dev(t), stats, dx2, False, dw, db, None)
    assert relerr(dx2, dx_ref) < TOL


@pytest.mark.parametrize("c1,c2,ncls,N,sp", [
    (200, 200, 2, 1, (10, 37, 37)),        # neuro3d_lite@183's tail: 214 tiles of 64 positions
    (200, 200, 2, 1, (5, 21, 21)),         # neuro3d@185's: 2,205 positions -> rows split 4 ways
    (150, 200, 2, 1, (4, 30, 31)),         # 3,720 positions -> tiles of 32, rows split 2 ways
    (37, 53, 3, 2, (3, 7, 13)),            # odd channel counts, batch, partial tiles
    (208, 208, 4, 1, (2, 9, 11)),          # the largest channel counts, 4 classes
    (200, 24, 2, 3, (1, 5, 7)),            # few positions
])
def test_fused_tail(ctx, c1, c2, ncls, N, sp):
    """csrc/tail.hip: [1x1x1 conv + bias + relu] -> [1x1x1 'lin' conv -> softmax ->
    MultinoulliNLL], forward and backward in ONE launch + its slot reduction: probabilities,
    loss, the head's dW / dbias, the 1x1x1 layer's pre-activation gradient (its weight gradient's
    operand) and bias gradient, and the data gradient, against the float64 oracle; unlabelled
    voxels, relu units at exactly zero (slope 0.5), all three wave layouts."""
    rng = np.random.RandomState(c1 + c2 + ncls)
    k = (1, 1, 1)
    x = rng.rand(N, c1, *sp).astype(np.float32)
    w1 = (rng.randn(c2, c1, *k) / np.sqrt(c1)).astype(np.float32)
    b1 = (rng.randn(c2) / 4).astype(np.float32)
    # a unit whose pre-activation is EXACTLY zero at every position: weights 0, bias 0
    w1[3] = 0.0; b1[3] = 0.0
    wh = (rng.randn(ncls, c2, *k) / np.sqrt(c2)).astype(np.float32)
    bh = (rng.randn(ncls) / 4).astype(np.float32)
    t = rng.randint(-1, ncls, (N, 1) + sp).astype(np.float32)
    pre = O.conv3d_fwd(x, w1)
    h = O.bias_act_fwd(pre, b1, 'relu')
    logits = O.conv3d_fwd(h, wh) + bh.reshape(1, -1, 1, 1, 1)
    loss_ref, dlog, p_ref = O.nll_loss_and_grad(logits, t)
    dwh_ref = O.conv3d_wgrad(dlog, h, wh.shape)
    dbh_ref = dlog.sum(axis=(0, 2, 3, 4))
    dh = O.conv3d_dgrad(dlog, wh, h.shape)
    dpre_ref, db1_ref = O.bias_act_bwd(dh, pre, b1, 'relu')
    dx_ref = O.conv3d_dgrad(dpre_ref, w1, x.shape)
    assert ctx.tail_supported(c1, c2, ncls)
    wpf = torch.zeros(ctx.conv_ws_bytes(c2, c1, k) // 4 + 64, device="cuda")
    wpd = torch.zeros_like(wpf)
    ctx.conv3d_pack(dev(w1), 0, wpf)
    ctx.conv3d_pack(dev(w1), 1, wpd)
    probs = torch.full((N, ncls) + sp, float("nan"), device="cuda")
    dpre = torch.full((N, c2) + sp, float("nan"), device="cuda")
    dx = torch.full((N, c1) + sp, float("nan"), device="cuda")
    stats = torch.full((2,), float("nan"), device="cuda")
    ws = torch.full((ctx.tail_ws_bytes(x.shape, c2, ncls) // 4 + 16,), float("nan"), device="cuda")
    whd = dev(wh.reshape(ncls, c2))
    ns = ctx.tail_fwd_bwd(dev(x), wpf, wpd, dev(b1), whd, dev(bh), dev(t), probs, dpre, dx,
                          stats, ws)
    dwh = torch.full((ncls, c2), 1.0, device="cuda")
    dbh = torch.full((ncls,), 2.0, device="cuda")
    db1 = torch.full((c2,), 3.0, device="cuda")
    loss = torch.zeros(1, device="cuda")
    ctx.tail_reduce(ws, ns, c2, ncls, dwh, dbh, db1, stats, loss)
    assert relerr(probs, p_ref) < TOL
    assert abs(float(stats[1]) - (t >= 0).sum()) < 0.5
    assert abs(float(loss) - loss_ref) / loss_ref < 1e-5
    assert relerr(dpre, dpre_ref) < TOL
    assert relerr(dx, dx_ref) < TOL
    assert relerr(dwh - 1.0, dwh_ref.reshape(ncls, c2)) < 1e-4       # ADDED to what was there
    assert relerr(dbh - 2.0, dbh_ref) < 1e-4
    assert relerr(db1 - 3.0, db1_ref) < 1e-4
    assert np.abs(dpre_ref[:, 3]).max() > 0                          # the slope-0.5 unit is live
    # dx THROUGH the activation backward of the layer that produced x, into the interior of that
    # layer's zero-padded gradient buffer (row pitch W + 2, two border planes), its bias gradient
    # from the slots: relu slope off the activated output (mode 1: signed zeros), off the
    # pre-activation + bias (mode 2), linear (mode 3)
    dpre_first = dpre.clone()
    pre0 = rng.randn(N, c1, *sp).astype(np.float32)
    pre0[:, 1] = 0.0                                             # units at exactly zero
    b0 = (rng.randn(c1) / 4).astype(np.float32); b0[1] = 0.0
    x0 = O.bias_act_fwd(pre0, b0, 'relu')                        # what the layer before emits
    signed = np.where(pre0 + b0.reshape(1, -1, 1, 1, 1) < 0, np.float32(-0.0), x0).astype(np.float32)
    pre_b = O.conv3d_fwd(x0, w1)
    h_b = O.bias_act_fwd(pre_b, b1, 'relu')
    lg_b = O.conv3d_fwd(h_b, wh) + bh.reshape(1, -1, 1, 1, 1)
    _, dlog_b, _ = O.nll_loss_and_grad(lg_b, t)
    dpre_b, _ = O.bias_act_bwd(O.conv3d_dgrad(dlog_b, wh, h_b.shape), pre_b, b1, 'relu')
    dx_b = O.conv3d_dgrad(dpre_b, w1, x0.shape)
    for mode in (1, 2, 3):
        dy0_ref, db0_ref = O.bias_act_bwd(dx_b, pre0, b0, 'lin' if mode == 3 else 'relu')
        padded = torch.zeros((N, c1, sp[0] + 2, sp[1], sp[2] + 2), device="cuda")
        view = padded[:, :, 1:-1, :, 1:-1]
        src = {1: dev(signed), 2: dev(pre0), 3: None}[mode]
        ns3 = ctx.tail_fwd_bwd(dev(x0), wpf, wpd, dev(b1), whd, dev(bh), dev(t), probs, dpre, view,
                               stats, ws, gm_mode=mode, gm_src=src,
                               gm_bias=dev(b0) if mode == 2 else None)
        dbp = torch.full((c1,), 5.0, device="cuda")
        dwh.zero_(); dbh.zero_(); db1.zero_()
        ctx.tail_reduce(ws, ns3, c2, ncls, dwh, dbh, db1, stats, loss, db_parent=dbp)
        assert relerr(view, dy0_ref) < TOL, mode
        assert relerr(dbp - 5.0, db0_ref) < 1e-4, mode
        assert float(padded[:, :, 0].abs().max()) == 0.0 and float(padded[..., 0].abs().max()) == 0.0
        assert relerr(dpre, dpre_b) < TOL
    # no data gradient wanted (the 1x1x1 layer directly on an input): dx = None
    dpre2 = torch.full_like(dpre, float("nan"))
    ns2 = ctx.tail_fwd_bwd(dev(x), wpf, None, dev(b1), whd, dev(bh), dev(t), probs, dpre2, None,
                           stats, ws)
    assert ns2 == ns and torch.equal(dpre2, dpre_first)


def test_pack_multi_equals_single_pack(ctx):
    """e2_conv3d_pack_multi rewrites only the weight-carrying part of the (zero-filled)
    images; the result must equal e2_conv3d_pack's full image, in both modes, also after a
    second call with changed weights (padding untouched)."""
    rng = np.random.RandomState(5)
    shapes = [(37, 10, 1, 3, 3), (200, 150, 1, 3, 3), (20, 1, 1, 4, 4), (40, 21, 3, 3, 3),
              (2, 200, 1, 1, 1)]
    ws = [dev(rng.randn(*s)) for s in shapes]
    jobs, singles = [], []
    for w in ws:
        co, ci = w.shape[:2]
        k = tuple(w.shape[2:])
        n = ctx.conv_ws_bytes(co, ci, k) // 4 + 64
        for mode in (0, 1):
            img = torch.zeros(n, device="cuda")
            ref = torch.full((n,), float("nan"), device="cuda")
            jobs.append((w, img, mode))
            singles.append((w, ref, mode))
    jd, nj = ctx.make_pack_jobs(jobs)
    for rep in range(2):
        ctx.conv3d_pack_multi(jd, nj)
        for (w, img, mode), (_, ref, _) in zip(jobs, singles):
            ctx.conv3d_pack(w, mode, ref)
            n_img = int(torch.isfinite(ref).sum().item())     # the part e2_conv3d_pack wrote
            assert n_img > w.numel()
            assert torch.equal(img[:n_img], ref[:n_img])
            ref.fill_(float("nan"))
        for w in ws:
            w.mul_(-0.5)


def test_pack_multi_with_row_hints(ctx):
    """e2_pack_job_set_rows: the repack told how far the reading launch's M tiles reach rewrites
    the real rows + the padding rows up to there; the image still equals e2_conv3d_pack's full
    image everywhere (rows beyond the hint are the zeros of the one-time fill, which is what the
    full image holds there too), and fewer elements are touched than with the default."""
    rng = np.random.RandomState(6)
    shapes = [(40, 20, 3, 3, 3), (200, 150, 1, 3, 3), (100, 80, 3, 4, 4)]
    hints = {(40, 0): 48, (40, 1): 20, (200, 0): 208, (200, 1): 160, (100, 0): 112, (100, 1): 80}
    ws = [dev(rng.randn(*s)) for s in shapes]
    jobs, rows = [], []
    for w in ws:
        co, ci = w.shape[:2]
        n = ctx.conv_ws_bytes(co, ci, tuple(w.shape[2:])) // 4 + 64
        for mode in (0, 1):
            jobs.append((w, torch.zeros(n, device="cuda"), mode))
            rows.append(hints[(co, mode)])
    poison = [(w, torch.full_like(img, float("nan")), mode) for (w, img, mode) in jobs]
    ctx.conv3d_pack_multi(*ctx.make_pack_jobs(jobs, rows))
    ctx.conv3d_pack_multi(*ctx.make_pack_jobs(poison, rows))          # which elements are written at all
    dflt = [(w, torch.full_like(img, float("nan")), mode) for (w, img, mode) in jobs]
    ctx.conv3d_pack_multi(*ctx.make_pack_jobs(dflt))
    for (w, img, mode), (_, pz, _), (_, df, _) in zip(jobs, poison, dflt):
        ref = torch.full_like(img, float("nan"))
        ctx.conv3d_pack(w, mode, ref)
        n_img = int(torch.isfinite(ref).sum().item())
        assert torch.equal(img[:n_img], ref[:n_img])
        written, written_default = int(torch.isfinite(pz).sum()), int(torch.isfinite(df).sum())
        assert w.numel() <= written < written_default, (written, written_default)


@pytest.mark.parametrize("ci,co,k,sp,tile,rows", [(150, 200, (1, 3, 3), (2, 11, 25), "7,2,32,1", 240),
                                                 (40, 150, (2, 4, 4), (3, 12, 13), "10,2,16,2", 176),
                                                 (20, 40, (3, 3, 3), (5, 14, 19), "3,4,12,1", 64),
                                                 (200, 200, (1, 1, 1), (2, 9, 20), "1,13,1", 224)])
def test_packed_images_with_their_own_row_length(ctx, ci, co, k, sp, tile, rows):
    """e2_set_image_rows / e2_pack_job_set_stride (DESIGN finding 52): an image packed with rows as
    long as its launch's tiles reach instead of the any-tiling formula -- forward, fused bias +
    relu, and data gradient give the formula's results; the launch must be told the length the
    image was packed with, and a tiling that reaches past it is an ERROR, never an over-read."""
    from elektronn2_amd.backend import E2Error
    rng = np.random.RandomState(ci + co)
    x = rng.rand(1, ci, *sp).astype(np.float32)
    w = (rng.randn(co, ci, *k) / np.sqrt(ci * np.prod(k))).astype(np.float32)
    b = (rng.randn(co) / 4).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    out_ref, _ = O.conv_node_fwd(x, w, b, (1, 1, 1), 'relu')
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    dx_ref = O.conv3d_dgrad(dy, w, x.shape)
    wd, xd = dev(w), dev(x)
    n = ctx.conv_ws_bytes(co, ci, k) // 4 + 64
    imgs = {0: torch.zeros(n, device="cuda"), 1: torch.zeros(n, device="cuda")}
    rows_d = -(-ci // 16) * 16 + 16 if tile.count(",") == 3 else 0       # the dgrad image: M = ci
    ctx.conv3d_pack_multi(*ctx.make_pack_jobs([(wd, imgs[0], 0), (wd, imgs[1], 1)], strides=[rows, rows_d]))
    pshape = (1, co) + tuple(y_ref.shape[2 + i] + 2 * (k[i] - 1) for i in range(3))
    dyp = torch.zeros(pshape, device="cuda")
    dyp[:, :, k[0] - 1:k[0] - 1 + y_ref.shape[2], k[1] - 1:k[1] - 1 + y_ref.shape[3],
        k[2] - 1:k[2] - 1 + y_ref.shape[4]] = dev(dy)
    ctx.set_tiling("igemm", tile)
    try:
        y = torch.full(y_ref.shape, float("nan"), device="cuda")
        ctx.set_image_rows(rows)
        ctx.conv3d_fwd_packed(xd, imgs[0], co, k, y)
        assert relerr(y, y_ref) < TOL
        if k[2] in (1, 3, 4, 5) and tile.count(",") == 3:
            out = torch.full(y_ref.shape, float("nan"), device="cuda")
            ctx.conv3d_fwd_packed_act(xd, imgs[0], co, k, dev(b), 'relu', out)
            assert relerr(out, out_ref) < TOL
        # the single-image pack under the announced length writes the same image
        again = torch.zeros(n, device="cuda")
        ctx.conv3d_pack(wd, 0, again)
        assert torch.equal(again, imgs[0])
        # told the formula instead, the launch would read the rows at the wrong pitch: results differ
        ctx.set_image_rows(0)
        ctx.conv3d_fwd_packed(xd, imgs[0], co, k, y)
        assert not relerr(y, y_ref) < TOL
    finally:
        ctx.set_image_rows(0)
        ctx.set_tiling("igemm", None)
    if rows_d:
        nb = -(-ci // 16)
        mt = [m for m in (5, 4, 3, 2, 1) if -(-nb // m) * m * 16 <= rows_d][0]     # (its tiles fit the rows)
        ctx.set_tiling("igemm", "%d,2,16,1" % mt)
        try:
            dx = torch.full(x.shape, float("nan"), device="cuda")
            ctx.set_image_rows(rows_d)
            ctx.conv3d_dgrad_packed(dyp, imgs[1], ci, k, dx)
            assert relerr(dx, dx_ref) < TOL
        finally:
            ctx.set_image_rows(0)
            ctx.set_tiling("igemm", None)
    # a tiling whose tiles reach past the announced rows: refused
    if tile.count(",") == 3 and co > 16 * 10 + 16:
        ctx.set_tiling("igemm", "10,2,16,1")               # 2 x 160 rows for 200 channels
        try:
            ctx.set_image_rows(rows)
            with pytest.raises(E2Error):
                ctx.conv3d_fwd_packed(xd, imgs[0], co, k, torch.empty(y_ref.shape, device="cuda"))
        finally:
            ctx.set_image_rows(0)
            ctx.set_tiling("igemm", None)


def test_fill_multi_and_skip_zero_fill(ctx):
    """e2_fill_multi zeroes many regions in one launch; a split-K conv reports the region it
    zero-filled (e2_conv_last_zero_fill) and, told that the output is already zero
    (e2_set_skip_zero_fill), accumulates onto it without its own fill"""
    a = torch.full((1000,), 3.0, device="cuda"); b = torch.full((77,), 4.0, device="cuda")
    c = torch.full((5,), 5.0, device="cuda")
    ptrs = torch.tensor([a.data_ptr(), b[7:].data_ptr()], dtype=torch.int64, device="cuda")
    cnts = torch.tensor([1000, 60], dtype=torch.int64, device="cuda")
    ctx.fill_multi(ptrs, cnts, 2, 0.0)
    assert float(a.abs().sum()) == 0 and float(b[7:67].abs().sum()) == 0
    assert float(b[:7].sum()) == 28 and float(b[67:].sum()) == 40 and float(c.sum()) == 25

    rng = np.random.RandomState(9)
    x = rng.rand(1, 24, 4, 9, 11).astype(np.float32)
    w = (rng.randn(20, 24, 2, 3, 3) / 10).astype(np.float32)
    ref = O.conv3d_fwd(x, w)
    ws = torch.empty(ctx.conv_ws_bytes(20, 24, (2, 3, 3)) // 4 + 64, device="cuda")
    ctx.conv3d_pack(dev(w), 0, ws)
    ctx.set_tiling("igemm", "2,1,8,4")                 # split-K 4: atomics onto a zeroed output
    try:
        y = torch.full(ref.shape, float("nan"), device="cuda")
        ctx.conv3d_fwd_packed(dev(x), ws, 20, (2, 3, 3), y)
        ptr, n = ctx.conv_last_zero_fill()
        assert ptr == y.data_ptr() and n == y.numel()
        assert relerr(y, ref) < TOL
        ctx.set_skip_zero_fill(True)
        try:
            y.fill_(1.0)                               # "already zero" is the caller's promise:
            ctx.conv3d_fwd_packed(dev(x), ws, 20, (2, 3, 3), y)   # ... the launch only adds
        finally:
            ctx.set_skip_zero_fill(False)
        assert relerr(y - 1.0, ref) < 1e-4
        ctx.set_tiling("igemm", "2,1,8,1")             # no split-K: nothing to zero
        ctx.conv3d_fwd_packed(dev(x), ws, 20, (2, 3, 3), y)
        assert ctx.conv_last_zero_fill()[1] == 0
        assert relerr(y, ref) < TOL
    finally:
        ctx.set_tiling("igemm", None)


def _random_conv_problems(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        ci = int(rng.choice([1, 2, 3, 5, 8, 17, 20, 31, 40, 64, 100, 150, 200, 257]))
        co = int(rng.choice([1, 2, 4, 7, 16, 20, 33, 40, 80, 150, 200, 208, 259]))
        k = tuple(int(v) for v in rng.choice([1, 2, 3, 4, 5], 3))
        sp = tuple(int(k[i] + rng.randint(0, 9 if i else 3)) for i in range(3))
        nb = int(rng.choice([1, 1, 2, 3]))
        if ci * co * np.prod(k) * np.prod(sp) * nb > 3e8:
            continue
        out.append((nb, ci, co, k, sp))
    return out


@pytest.mark.parametrize("case", _random_conv_problems(24, 2026), ids=lambda c: "n%d_%d-%d_k%dx%dx%d_s%dx%dx%d" % (
    (c[0], c[1], c[2]) + c[3] + c[4]))
def test_conv3d_random_shapes_with_the_librarys_own_tilings(ctx, case):
    """drop-in means any shape a caller brings: odd channel counts (1 .. 259), every kernel
    extent 1 .. 5 (specialised and generic tap rows), outputs as small as one voxel, batch
    1 .. 3 -- forward, data gradient and the padded-buffer weight gradient with NO tiling
    forced and nothing tuned, against the f64 oracle"""
    N, Ci, Co, k, sp = case
    rng = np.random.RandomState(Ci * 7 + Co)
    x = rng.rand(N, Ci, *sp).astype(np.float32)
    w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
    y_ref = O.conv3d_fwd(x, w)
    y = torch.full(y_ref.shape, float("nan"), device="cuda")
    ctx.conv3d_fwd(dev(x), dev(w), y)
    assert relerr(y, y_ref) < TOL
    dy = rng.randn(*y_ref.shape).astype(np.float32)
    osp = y_ref.shape[2:]
    pad = [kk - 1 for kk in k]
    pshape = (N, Co) + tuple(osp[i] + 2 * pad[i] for i in range(3))
    flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
    dyp = flat[:int(np.prod(pshape))].view(pshape)
    dyp[:, :, pad[0]:pad[0] + osp[0], pad[1]:pad[1] + osp[1], pad[2]:pad[2] + osp[2]] = dev(dy)
    dx = torch.full(x.shape, float("nan"), device="cuda")
    ctx.conv3d_dgrad(dyp, dev(w), dx)
    assert relerr(dx, O.conv3d_dgrad(dy, w, x.shape)) < TOL
    dw = torch.full(w.shape, float("nan"), device="cuda")
    ctx.conv3d_wgrad_pad(dev(x), dyp, dw)
    assert relerr(dw, O.conv3d_wgrad(dy, x, w.shape)) < TOL


def test_step_prologue_follows_the_launch_count(ctx):
    """e2_step_prologue (several steps in one graph, DESIGN finding 55): launch number L copies
    ring slot L % n_slots -- bit-exact, for slot sizes with and without a float4 tail, one
    work-group and the grid's cap -- and stores what the step before it left in `src` in history
    slot (L - 1) % n; the same launches replayed from a captured graph keep counting."""
    dev = ctx.device
    rng = np.random.RandomState(3)
    for n_slots, floats in ((3, 8), (4, 1000), (2, 4 * 256 * 1024 + 4 * 77)):
        ring = torch.tensor(rng.rand(n_slots, floats).astype(np.float32), device=dev)
        dst = torch.zeros(floats + 8, device=dev)
        state = torch.zeros(2, dtype=torch.int64, device=dev)
        for L in range(2 * n_slots + 1):
            ctx.step_prologue(state, ring=ring, dst=dst)
            assert torch.equal(dst[:floats], ring[L % n_slots]), (n_slots, floats, L)
            assert float(dst[floats:].abs().max()) == 0.0
        assert state.cpu().tolist() == [2 * n_slots + 1, 0]
    # history alone: 5 values per step, 4 slots; launch L stores what "step L - 1" left in src
    hist = torch.zeros(4, 5, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    src = torch.full((8,), -1.0, device=dev)
    for L in range(7):
        ctx.step_prologue(state, src=src, hist=hist)
        src[:5] = torch.arange(5, device=dev) + 10.0 * L          # "the loss of step L"
    torch.cuda.synchronize()
    assert state.cpu().tolist() == [7, 0]
    want = np.array([[40, 41, 42, 43, 44], [50, 51, 52, 53, 54], [20, 21, 22, 23, 24], [30, 31, 32, 33, 34]], np.float32)
    assert np.array_equal(hist.cpu().numpy(), want)
    # both, inside a captured graph: three "steps" per launch, two launches
    ring = torch.tensor(rng.rand(4, 64).astype(np.float32), device=dev)
    dst = torch.zeros(64, device=dev)
    loss = torch.zeros(4, device=dev)
    hist = torch.zeros(8, 1, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    s = torch.cuda.Stream(device=dev)
    old = ctx.stream
    torch.cuda.synchronize()
    ctx.set_stream(s)
    try:
        with torch.cuda.stream(s):
            ctx.graph_begin()
            for _ in range(3):
                ctx.step_prologue(state, ring=ring, dst=dst, src=loss, hist=hist)
                ctx.copy5(dst[:1].view(1, 1, 1, 1, 1), loss[:1].view(1, 1, 1, 1, 1))   # "the step": loss = batch[0]
            g = ctx.graph_end()
            ctx.graph_launch(g)
            ctx.graph_launch(g)
        s.synchronize()
        ctx.graph_destroy(g)
    finally:
        ctx.set_stream(old)
    got = hist.cpu().numpy()[:, 0]
    for L in range(5):
        assert got[L] == float(ring[L % 4, 0]), L
    assert float(loss[0]) == float(ring[5 % 4, 0]) and float(hist[5:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ctx.step_prologue(state, ring=torch.zeros(2, 6, device=dev), dst=torch.zeros(8, device=dev))   # 24-byte slots
