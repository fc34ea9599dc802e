"""bf16 operand form of the conv GEMMs (e2_set_mfma_dtype): operands
rounded to bf16 (nearest even) on their way into the matrix core, f32 accumulation,
tensors in memory stay f32.

The checker is the f64 oracle fed with operands that were rounded to bf16 beforehand:
against it the kernels are held to the same 2e-5 as the f32 path (only the summation
order differs).  Against the unrounded oracle the error must be of bf16 size -- above
1e-4 (the bf16 form really ran) and below 2e-2.  Separate, looser end-to-end tolerance
for a training step."""
import os

import numpy as np
import pytest
import torch

from oracle import e2_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


def bf16_round(a):
    return torch.tensor(np.asarray(a, np.float32)).bfloat16().float().numpy()


def dev(a):
    return torch.tensor(np.asarray(a, np.float32), device="cuda")


def relerr(got, ref):
    got = got.detach().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.fixture()
def bctx(ctx):
    ctx.set_mfma_dtype('bf16')
    assert ctx.mfma_dtype == 'bf16'
    yield ctx
    ctx.set_mfma_dtype('f32')


CASES = [
    # Cin, Cout, k, in spatial, igemm force, wgrad force
    (20, 40, (3, 3, 3), (5, 14, 19), None, None),
    (40, 150, (2, 4, 4), (3, 12, 13), None, None),
    (150, 200, (1, 3, 3), (2, 11, 12), "7,2,32,1", "7,2,1,256,4"),
    (200, 200, (1, 1, 1), (2, 9, 10), "10,1,16,1", None),          # GU = 4
    (200, 200, (1, 1, 1), (2, 9, 10), "7,2,8,2", "4,1,1,256,2"),   # GU = 1, split-K
    (20, 30, (1, 5, 5), (2, 17, 18), None, "2,2,14,256,3"),        # 16x16x32 form, waves split quads
    (30, 40, (1, 5, 5), (1, 47, 47), "3,4,16,1", None),
    (24, 200, (3, 2, 3), (5, 13, 37), "4,4,24,1", "4,4,1,256,6"),
    (150, 200, (1, 3, 3), (2, 11, 12), None, "7,2,0,128,3"),        # LDS-staged wgrad kernel
    (40, 150, (2, 4, 4), (3, 12, 13), None, "5,4,0,64,4"),
    (6, 17, (1, 1, 3), (1, 1, 70), None, None),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%d-%d_k%s_%s_%s" % (
    c[0], c[1], "x".join(map(str, c[2])), c[4], c[5]))
def test_fwd_dgrad_wgrad_bf16(bctx, case):
    Ci, Co, k, sp, fi, fw = case
    rng = np.random.RandomState(abs(hash(case[:4])) % (2 ** 31))
    x = rng.rand(1, Ci, *sp).astype(np.float32)
    w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
    xb, wb = bf16_round(x), bf16_round(w)
    y_ref, y_f32 = O.conv3d_fwd(xb, wb), O.conv3d_fwd(x, w)
    xd, wd = dev(x), dev(w)
    ws = torch.empty(bctx.conv_ws_bytes(Co, Ci, k) // 4 + 64, device="cuda")
    bctx.set_tiling("igemm", fi or None)
    try:
        bctx.conv3d_pack(wd, 0, ws)
        y = torch.full(y_ref.shape, float("nan"), device="cuda")
        bctx.conv3d_fwd_packed(xd, ws, Co, k, y)
        e_b, e_f = relerr(y, y_ref), relerr(y, y_f32)
        assert e_b < TOL, e_b
        assert 1e-4 < e_f < 2e-2, e_f

        dy = rng.randn(*y_ref.shape).astype(np.float32)
        osp = y_ref.shape[2:]
        pshape = (1, Co) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
        flat = torch.zeros(int(np.prod(pshape)) + 32, device="cuda")
        dyp = flat[:int(np.prod(pshape))].view(pshape)
        inner = dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1],
                    k[2] - 1:k[2] - 1 + osp[2]]
        inner.copy_(dev(dy))
        bctx.conv3d_pack(wd, 1, ws)
        dx = torch.full(x.shape, float("nan"), device="cuda")
        bctx.conv3d_dgrad_packed(dyp, ws, Ci, k, dx)
        dyb = bf16_round(dy)
        assert relerr(dx, O.conv3d_dgrad(dyb, wb, x.shape)) < TOL
        assert 1e-4 < relerr(dx, O.conv3d_dgrad(dy, w, x.shape)) < 2e-2
    finally:
        bctx.set_tiling("igemm", None)
    bctx.set_tiling("wgrad", fw or None)
    try:
        dw = torch.full(w.shape, float("nan"), device="cuda")
        bctx.conv3d_wgrad_pad(xd, dyp, dw)
    finally:
        bctx.set_tiling("wgrad", None)
    assert relerr(dw, O.conv3d_wgrad(dyb, xb, w.shape)) < TOL
    assert 1e-4 < relerr(dw, O.conv3d_wgrad(dy, x, w.shape)) < 2e-2


def test_fused_bias_act_epilogue_bf16(bctx):
    rng = np.random.RandomState(3)
    x = rng.rand(1, 12, 3, 12, 22).astype(np.float32)
    w = (rng.randn(37, 12, 1, 3, 3) / 6).astype(np.float32)
    b = (rng.randn(37) / 4).astype(np.float32)
    ref, _ = O.conv_node_fwd(bf16_round(x), bf16_round(w), b, (1, 1, 1), 'relu')
    ws = torch.empty(bctx.conv_ws_bytes(37, 12, (1, 3, 3)) // 4 + 64, device="cuda")
    bctx.conv3d_pack(dev(w), 0, ws)
    y = torch.full(ref.shape, float("nan"), device="cuda")
    bctx.conv3d_fwd_packed_act(dev(x), ws, 37, (1, 3, 3), dev(b), 'relu', y)
    assert relerr(y, ref) < TOL


@pytest.fixture()
def process_bf16():
    """the launch plans run on the process-wide context"""
    import elektronn2_amd
    elektronn2_amd.set_mfma_dtype('bf16')
    yield
    elektronn2_amd.set_mfma_dtype('f32')


def test_training_step_bf16_close_to_f32_oracle(process_bf16):
    """neuro3d_lite, one step of gradients in bf16 arithmetic against the f64 oracle of
    the f32 net: loss within 1e-2, every gradient tensor within 1e-1 of its max (bf16
    keeps 8 bits; the first layer's gradient has passed seven rounded layers twice) and
    pointing the same way (cosine > 0.995); then Adam steps stay finite and learn."""
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    spec, sp = O.NEURO3D_LITE, (7, 47, 47)
    params = O.init_net(spec, 1, seed=1)
    rng = np.random.RandomState(0)
    x = rng.rand(1, 1, *sp).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    m = nets.neuro3d_lite((None, 1) + sp, params=params)
    m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    loss_ref, grads_ref, _ = O.net_loss_and_grads(spec, params, x, t)
    loss = float(m.loss(x, t))
    assert abs(loss - loss_ref) < 1e-2 * abs(loss_ref)
    assert abs(loss - loss_ref) > 1e-7 * abs(loss_ref)        # not the f32 path
    got = m.gradients(x, t)
    flat_ref = []
    for gw, gb in grads_ref:
        flat_ref += [gw, gb]
    for g in got:
        cands = [r for r in flat_ref if r.shape == g.shape]
        best = min(cands, key=lambda r: np.abs(g - r).max())
        assert np.abs(g - best).max() < 1e-1 * np.abs(best).max()
        cos = float((g * best).sum() / np.sqrt((g * g).sum() * (best * best).sum() + 1e-30))
        assert cos > 0.995, cos
    losses = [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(30)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


@pytest.mark.parametrize("tile", [None, "32,1,1", "32,1,2", "32,2,1", "32,4,1", "32,4,2", "32,2,4", "32,1,4"])
@pytest.mark.parametrize("case", [(20, 40, (3, 3, 3), (2, 5, 14, 19)), (150, 200, (1, 3, 3), (1, 2, 11, 12)),
                                  (40, 150, (2, 4, 4), (1, 3, 12, 13)), (200, 70, (1, 1, 1), (2, 2, 9, 10)),
                                  (33, 17, (1, 5, 2), (1, 1, 9, 70))])
def test_conv_with_bf16_operands_in_memory(ctx, case, tile):
    """csrc/conv_bf16.hip (e2_conv3d_fwd_bf16 / e2_conv3d_dgrad_bf16): bf16 planes of 16-byte
    pixels, input windows staged in LDS once per kernel plane, filter rows from L2, 32x32x16
    MFMA.  Same bounds as the operand-rounding form: 2e-5 against the f64 oracle on
    bf16-rounded operands; fused bias + relu with signed zeros; strided output view (the
    interior of a padded buffer); every wave tile; kd = 1 / 2 / 3 (single and double
    buffered windows), ragged channel counts, step counts that need zero padding."""
    Ci, Co, k, (N, D, H, W) = case
    rng = np.random.RandomState(Ci + Co)
    x = rng.rand(N, Ci, D, H, W).astype(np.float32)
    w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
    b = (rng.randn(Co) * 0.1).astype(np.float32)
    xr, wr = bf16_round(x), bf16_round(w)
    y_ref = O.conv3d_fwd(xr, wr)
    ctx.set_tiling("igemm", tile)
    try:
        y = torch.full(y_ref.shape, float("nan"), device="cuda")
        ctx.conv3d_fwd_bf16(dev(x), dev(w), y)
        assert relerr(y, y_ref) < TOL
        # a strided input view (a crop of a larger buffer, channel slice of a concat buffer)
        host = torch.full((N, Ci + 3, D + 1, H + 2, W + 5), float("nan"), device="cuda")
        xv = host[:, 2:2 + Ci, 1:, 1:-1, 3:-2]
        xv.copy_(dev(x))
        y.fill_(float("nan"))
        ctx.conv3d_fwd_bf16(xv, dev(w), y)
        assert relerr(y, y_ref) < TOL
        e = relerr(y, O.conv3d_fwd(x, w))
        assert 1e-4 < e < 2e-2, e                       # bf16 really ran
        # fused bias + relu into a strided view
        big = torch.zeros((N, Co, y_ref.shape[2] + 1, y_ref.shape[3] + 2, y_ref.shape[4] + 3), device="cuda")
        yv = big[:, :, 1:, 1:-1, 2:-1]
        ctx.conv3d_fwd_bf16(dev(x), dev(w), yv, bias=dev(b), act='relu')
        pre = y_ref + b.reshape(1, -1, 1, 1, 1)
        assert relerr(yv, np.maximum(pre, 0)) < TOL
        neg = torch.signbit(yv).cpu().numpy()
        assert neg[pre < -1e-5].all() and not neg[pre > 1e-5].any()
        # data gradient on the zero-padded gradient buffer
        dy = rng.randn(*y_ref.shape).astype(np.float32)
        osp = y_ref.shape[2:]
        pshape = (N, Co) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
        dyp = torch.zeros(pshape, device="cuda")
        dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1],
            k[2] - 1:k[2] - 1 + osp[2]] = dev(dy)
        dx = torch.full(x.shape, float("nan"), device="cuda")
        try:
            ctx.conv3d_dgrad_bf16(dyp, dev(w), dx)
        except Exception as err:      # the widest tile x most channels: refused, never wrong
            assert "does not fit LDS" in str(err) and tile in ("32,2,4", "32,1,4") and Co >= 150
            return
        assert relerr(dx, O.conv3d_dgrad(bf16_round(dy), wr, x.shape)) < TOL
    finally:
        ctx.set_tiling("igemm", None)




@pytest.mark.parametrize("tile", [None, "32,1,1,0,0", "32,1,2,0,3", "32,2,1,0,1", "32,2,2,0,7",
                                  "32,1,4,0,2", "32,2,3,0,5", "32,2,4,0,0",
                                  "32,1,1,1,0", "32,1,2,1,3", "32,2,1,1,1", "32,2,2,1,7",
                                  "32,1,4,1,2", "32,2,3,1,5", "32,2,4,1,0", "32,1,3,1,16"])
@pytest.mark.parametrize("case", [(20, 40, (3, 3, 3), (2, 5, 14, 19)), (150, 200, (1, 3, 3), (1, 2, 11, 12)),
                                  (40, 150, (2, 4, 4), (1, 3, 12, 13)), (200, 70, (1, 1, 1), (2, 2, 9, 10)),
                                  (33, 17, (1, 5, 2), (1, 1, 9, 70)), (70, 130, (1, 2, 5), (1, 1, 20, 21))])
def test_wgrad_with_bf16_operands_in_memory(ctx, case, tile):
    """csrc/wgrad_bf16.hip (e2_conv3d_wgrad_bf16): K = positions; dy rows by ds_read_b128, the
    tap-shifted input windows from channels-last LDS pixels by the transposed read
    ds_read_b64_tr_b16; 2e-5 against the f64 oracle on bf16-rounded operands; overwrite and
    accumulate; the gradient as a strided interior view of its padded buffer; channel counts
    that are not multiples of 8 / 32 / 128, kernel rows shorter and longer than the tap group;
    both wave layouts (fourth tiling field: 0 = four blocks of 32 input channels of one kernel
    row, 1 = 32 input channels of four kernel rows -- kernel planes of 1, 2, 3, 4 and 5 rows)."""
    Ci, Co, k, (N, D, H, W) = case
    rng = np.random.RandomState(Ci * 3 + Co)
    x = rng.rand(N, Ci, D, H, W).astype(np.float32)
    osp = (D - k[0] + 1, H - k[1] + 1, W - k[2] + 1)
    dy = rng.randn(N, Co, *osp).astype(np.float32)
    ref = O.conv3d_wgrad(bf16_round(dy), bf16_round(x), (Co, Ci) + tuple(k))
    pshape = (N, Co) + tuple(osp[i] + 2 * (k[i] - 1) for i in range(3))
    dyp = torch.zeros(pshape, device="cuda")
    dyv = dyp[:, :, k[0] - 1:k[0] - 1 + osp[0], k[1] - 1:k[1] - 1 + osp[1], k[2] - 1:k[2] - 1 + osp[2]]
    dyv.copy_(dev(dy))
    dw = torch.full((Co, Ci) + tuple(k), float("nan"), device="cuda")
    ctx.set_tiling("wgrad", tile)
    try:
        ctx.conv3d_wgrad_bf16(dev(x), dyv, dw)
        assert relerr(dw, ref) < TOL
        e = relerr(dw, O.conv3d_wgrad(dy, x, (Co, Ci) + tuple(k)))
        assert 1e-5 < e < 2e-2, e
        ctx.conv3d_wgrad_bf16(dev(x), dyv, dw, accumulate=True)
        assert relerr(dw, 2 * ref) < TOL
        # the forward's kept channels-last copy of x instead of a second conversion
        # (e2_conv3d_fwd_bf16_keep -> e2_conv3d_wgrad_bf16_xcl); the x handed to the weight
        # gradient is poisoned: only its shape may be used
        w = (rng.randn(Co, Ci, *k) / np.sqrt(Ci * np.prod(k))).astype(np.float32)
        keep = torch.zeros(ctx.conv_bf16_xkeep_bytes(x.shape, k), dtype=torch.uint8, device="cuda")
        y = torch.empty((N, Co) + osp, device="cuda")
        ctx.conv3d_fwd_bf16(dev(x), dev(w), y, xkeep=keep)
        assert relerr(y, O.conv3d_fwd(bf16_round(x), bf16_round(w))) < TOL
        poison = torch.full(x.shape, float("nan"), device="cuda")
        dw.fill_(float("nan"))
        ctx.conv3d_wgrad_bf16(poison, dyv, dw, xcl=keep)
        assert relerr(dw, ref) < TOL
    finally:
        ctx.set_tiling("wgrad", None)
