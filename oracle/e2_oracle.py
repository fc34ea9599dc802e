"""CPU ORACLE (test infrastructure, NOT product code).

NumPy float64 restatement of the arithmetic on ELEKTRONN2's 3-D conv / pool /
upconv training step.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``elektronn2_amd``) never does.

PARITY STATUS: **parity unpinned** for conv / pool / upconv / bias-act and
their gradients.  The reference delegates all arithmetic to Theano
(``>=0.8,<0.10``, ``setup.py:53``), which is neither vendored nor installed
and cannot be fetched (no network); ``import elektronn2`` fails here with
ordinary ``ModuleNotFoundError``s (numba / h5py / theano).  The reference's
own tests hold no golden numbers for this path (``tests/test_conv.py`` only
compares two Theano back ends on unseeded random data and needs CUDA).  The
single known-answer relation the reference states --
``conv1d(x, w) == np.convolve(x, w, 'valid')`` (``tests/test_conv.py:89-104``)
-- fixes the kernel-flip convention and IS checked (``tests/test_oracle.py``).
Independent second opinions used to pin the restatement: torch-CPU autograd
and ``scipy.signal.convolve`` (both in this container).

All citations are relative to /root/reference/elektronn2/.
Layout everywhere: activations (b, f, z, x, y) = NCDHW, weights
(n_f, n_in, kz, kx, ky) = KCDHW  (neuromancer/neural.py:615-623, axis_order 'dnn').
"""
from __future__ import annotations

import numpy as np

F64 = np.float64
EPS_NLL = 1e-5        # neuromancer/loss.py:30
EPS_ADAM = 1e-5       # neuromancer/optimiser.py:289


# --------------------------------------------------------------------------
# conv  (neuromancer/computations.py:364-428 ; Theano conv3d2d / dnn_conv3d,
#        border_mode='valid', filter_flip=True  => TRUE convolution, F1)
# --------------------------------------------------------------------------
def conv3d_fwd(x, w):
    """y[n,co,z,x,y] = sum_ci sum_k w[co,ci,kz,kx,ky] *
    x[n,ci,z+Kz-1-kz, x+Kx-1-kx, y+Ky-1-ky]   ('valid')."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    N, Ci, D, H, W = x.shape
    Co, Ci2, kd, kh, kw = w.shape
    assert Ci == Ci2
    Do, Ho, Wo = D - kd + 1, H - kh + 1, W - kw + 1
    y = np.zeros((N, Co, Do, Ho, Wo), F64)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                # tap position t = K-1-k
                xs = x[:, :, kd - 1 - a:kd - 1 - a + Do,
                       kh - 1 - b:kh - 1 - b + Ho,
                       kw - 1 - c:kw - 1 - c + Wo]
                y += np.einsum('oi,nizxy->nozxy', w[:, :, a, b, c], xs,
                               optimize=True)
    return y


def conv3d_dgrad(dy, w, x_shape):
    """Gradient of conv3d_fwd wrt x (what T.grad produces, model.py:182)."""
    dy = np.asarray(dy, F64)
    w = np.asarray(w, F64)
    Co, Ci, kd, kh, kw = w.shape
    N, _, Do, Ho, Wo = dy.shape
    dx = np.zeros(x_shape, F64)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                dx[:, :, kd - 1 - a:kd - 1 - a + Do,
                   kh - 1 - b:kh - 1 - b + Ho,
                   kw - 1 - c:kw - 1 - c + Wo] += np.einsum(
                       'oi,nozxy->nizxy', w[:, :, a, b, c], dy, optimize=True)
    return dx


def conv3d_wgrad(dy, x, w_shape):
    """Gradient of conv3d_fwd wrt w."""
    dy = np.asarray(dy, F64)
    x = np.asarray(x, F64)
    Co, Ci, kd, kh, kw = w_shape
    N, _, Do, Ho, Wo = dy.shape
    dw = np.zeros(w_shape, F64)
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                xs = x[:, :, kd - 1 - a:kd - 1 - a + Do,
                       kh - 1 - b:kh - 1 - b + Ho,
                       kw - 1 - c:kw - 1 - c + Wo]
                dw[:, :, a, b, c] = np.einsum('nozxy,nizxy->oi', dy, xs,
                                              optimize=True)
    return dw


# --------------------------------------------------------------------------
# max-pool  (computations.py:538-631: pool_2d(ignore_border=True) on (x,y),
#            then T.maximum over z-slices; stride == pool)
# --------------------------------------------------------------------------
def _windows(x, pool):
    N, C, D, H, W = x.shape
    pz, py, px = pool
    Do, Ho, Wo = D // pz, H // py, W // px
    xv = x[:, :, :Do * pz, :Ho * py, :Wo * px]
    return xv.reshape(N, C, Do, pz, Ho, py, Wo, px)


def maxpool3d_fwd(x, pool):
    x = np.asarray(x, F64)
    if tuple(pool) == (1, 1, 1):      # computations.py:569-570
        return x.copy()
    return _windows(x, pool).max(axis=(3, 5, 7))


def maxpool3d_bwd(dy, x, pool):
    """Tie rule of the Theano CPU path: EVERY element equal to the window max
    receives the full gradient (MaxPoolGrad on xy, T.maximum grad on z;
    SURVEY.md §8 a-3)."""
    x = np.asarray(x, F64)
    dy = np.asarray(dy, F64)
    if tuple(pool) == (1, 1, 1):
        return dy.copy()
    N, C, D, H, W = x.shape
    pz, py, px = pool
    xw = _windows(x, pool)
    m = xw.max(axis=(3, 5, 7), keepdims=True)
    g = (xw == m) * dy[:, :, :, None, :, None, :, None]
    dx = np.zeros_like(x)
    Do, Ho, Wo = D // pz, H // py, W // px
    dx[:, :, :Do * pz, :Ho * py, :Wo * px] = g.reshape(
        N, C, Do * pz, Ho * py, Wo * px)
    return dx


# --------------------------------------------------------------------------
# bias + activation (neural.py:705-712 ; computations.py:57-134)
# --------------------------------------------------------------------------
def bias_act_fwd(y, b, act='relu'):
    pre = np.asarray(y, F64) + np.asarray(b, F64)[None, :, None, None, None]
    if act == 'relu':                      # T.nnet.relu = 0.5*(x+|x|)
        return 0.5 * (pre + np.abs(pre))
    if act == 'lin':
        return pre
    if act == 'tanh':
        return np.tanh(pre)
    if act == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-pre))
    if act == 'abs':
        return np.abs(pre)
    raise NotImplementedError(act)


def bias_act_bwd(dout, y, b, act='relu'):
    """returns (dy, db).  relu'(0) = 0.5 (Theano grad of 0.5*(x+|x|))."""
    pre = np.asarray(y, F64) + np.asarray(b, F64)[None, :, None, None, None]
    dout = np.asarray(dout, F64)
    if act == 'relu':
        d = 0.5 * (1.0 + np.sign(pre))
    elif act == 'lin':
        d = np.ones_like(pre)
    elif act == 'tanh':
        d = 1.0 - np.tanh(pre) ** 2
    elif act == 'sigmoid':
        s = 1.0 / (1.0 + np.exp(-pre))
        d = s * (1 - s)
    elif act == 'abs':
        d = np.sign(pre)
    else:
        raise NotImplementedError(act)
    dy = dout * d
    return dy, dy.sum(axis=(0, 2, 3, 4))


# --------------------------------------------------------------------------
# Conv node = conv -> pool -> +bias -> act   (neural.py:662-712, F3)
# --------------------------------------------------------------------------
def conv_node_fwd(x, w, b, pool=(1, 1, 1), act='relu', rnd=None):
    """``rnd``: optional rounding of the GEMM's two operands (the bf16 row: operands
    rounded to bf16, accumulation and every tensor in memory unrounded); None = the reference's
    float arithmetic."""
    r = rnd if rnd is not None else (lambda a: a)
    c = conv3d_fwd(r(x), r(w))
    p = maxpool3d_fwd(c, pool)
    return bias_act_fwd(p, b, act), (c, p)


def conv_node_bwd(dout, x, w, b, cache, pool=(1, 1, 1), act='relu',
                  need_dx=True, rnd=None):
    r = rnd if rnd is not None else (lambda a: a)
    c, p = cache
    dp, db = bias_act_bwd(dout, p, b, act)
    dc = maxpool3d_bwd(dp, c, pool)
    dw = conv3d_wgrad(r(dc), r(x), w.shape)
    dx = conv3d_dgrad(r(dc), r(w), x.shape) if need_dx else None
    return dx, dw, db


# --------------------------------------------------------------------------
# UpConv (neural.py:989-1072; computations.py:216-255, 749-782)
# --------------------------------------------------------------------------
def unpooling_nd(x, pool):
    """computations.py:749-782: zero stuffing; value lands at index
    p-1, 2p-1, ... ; size s*p + p-1."""
    x = np.asarray(x, F64)
    N, C, D, H, W = x.shape
    pz, py, px = pool
    out = np.zeros((N, C, D * pz + pz - 1, H * py + py - 1, W * px + px - 1),
                   F64)
    out[:, :, pz - 1::pz, py - 1::py, px - 1::px] = x
    return out


def upconv3d_fwd_literal(x, w, pool):
    """The CPU branch exactly as written (neural.py:1013-1020):
    unpooling -> conv(...,'valid') with w_sh=[n_f, n_in, *pool]."""
    return conv3d_fwd(unpooling_nd(x, pool), w)


def upconv3d_fwd(x, w, pool):
    """Closed form (F2): y[n,co,p*i+r] = sum_ci w[co,ci,r] * x[n,ci,i]."""
    x = np.asarray(x, F64)
    w = np.asarray(w, F64)
    N, Ci, D, H, W = x.shape
    Co = w.shape[0]
    pz, py, px = pool
    y = np.zeros((N, Co, D * pz, H * py, W * px), F64)
    for a in range(pz):
        for b in range(py):
            for c in range(px):
                y[:, :, a::pz, b::py, c::px] = np.einsum(
                    'oi,nizxy->nozxy', w[:, :, a, b, c], x, optimize=True)
    return y


def upconv3d_dgrad(dy, w, pool):
    dy = np.asarray(dy, F64)
    w = np.asarray(w, F64)
    pz, py, px = pool
    N, Co, D, H, W = dy.shape
    dx = np.zeros((N, w.shape[1], D // pz, H // py, W // px), F64)
    for a in range(pz):
        for b in range(py):
            for c in range(px):
                dx += np.einsum('oi,nozxy->nizxy', w[:, :, a, b, c],
                                dy[:, :, a::pz, b::py, c::px], optimize=True)
    return dx


def upconv3d_wgrad(dy, x, pool):
    dy = np.asarray(dy, F64)
    x = np.asarray(x, F64)
    pz, py, px = pool
    dw = np.zeros((dy.shape[1], x.shape[1], pz, py, px), F64)
    for a in range(pz):
        for b in range(py):
            for c in range(px):
                dw[:, :, a, b, c] = np.einsum(
                    'nozxy,nizxy->oi', dy[:, :, a::pz, b::py, c::px], x,
                    optimize=True)
    return dw


# --------------------------------------------------------------------------
# Crop / Concat (neural.py:1152-1168 ; node_basic.py:1433-1440)
# --------------------------------------------------------------------------
def crop(x, crop_sizes):
    """symmetric spatial slice: crop c on each side of each spatial axis."""
    sl = [slice(None), slice(None)]
    for c in crop_sizes:
        sl.append(slice(c, -c if c else None))
    return np.asarray(x)[tuple(sl)]


def crop_bwd(dy, x_shape, crop_sizes):
    dx = np.zeros(x_shape, F64)
    sl = [slice(None), slice(None)]
    for c in crop_sizes:
        sl.append(slice(c, -c if c else None))
    dx[tuple(sl)] = dy
    return dx


# --------------------------------------------------------------------------
# softmax / NLL / aggregate / errors
# --------------------------------------------------------------------------
def softmax(x, axis=1):
    """computations.py:175-176."""
    x = np.asarray(x, F64)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def multinoulli_nll(pred, target):
    """loss.py:261-347 with target_is_sparse=True, n_indep=1, no weights /
    masks.  pred (N,C,...) probabilities; target (N,1,...) class ids
    (float or int; ids outside [0,C) are 'unlabelled').  Returns the
    element-wise nll of shape (N,1,...)."""
    pred = np.asarray(pred, F64)
    target = np.asarray(target)
    C = pred.shape[1]
    classes = np.arange(C).reshape((1, C) + (1,) * (pred.ndim - 2))
    onehot = (target == classes).astype(F64)
    # xlogy0(x, y) = x*log(y), 0 where x == 0
    nll_up = -np.where(onehot == 0, 0.0, onehot * np.log(pred + EPS_NLL))
    n_tot = onehot.sum()
    nll = nll_up * pred.size / (n_tot + EPS_NLL) / 1 / C
    return nll.sum(axis=1, keepdims=True)


def aggregate_loss(*losses, mixing_weights=None):
    """loss.py:1357-1363: mean of (mean of each loss array * weight)."""
    means = np.array([np.mean(l) for l in losses], F64)
    if mixing_weights is None:
        mixing_weights = np.ones(len(losses))
    return float(np.mean(means * np.asarray(mixing_weights, F64)))


def nll_loss_and_grad(logits, target):
    """softmax -> MultinoulliNLL -> AggregateLoss and d(loss)/d(logits)."""
    logits = np.asarray(logits, F64)
    p = softmax(logits, 1)
    C = p.shape[1]
    classes = np.arange(C).reshape((1, C) + (1,) * (p.ndim - 2))
    onehot = (np.asarray(target) == classes).astype(F64)
    n_tot = onehot.sum()
    n_elem = p.size / C                       # elements of the nll array
    scale = p.size / (n_tot + EPS_NLL) / C / n_elem
    loss = float((-onehot * np.log(p + EPS_NLL)).sum() * scale)
    dp = -onehot / (p + EPS_NLL) * scale
    dlogits = p * (dp - (dp * p).sum(axis=1, keepdims=True))
    return loss, dlogits, p


def classification_errors(pred, target):
    """loss.py Errors (target_is_sparse=True): mean(argmax(pred) != target)
    over all voxels."""
    cls = np.argmax(pred, axis=1)
    t = np.asarray(target)[:, 0]
    return float(np.mean(cls != t))


# --------------------------------------------------------------------------
# optimisers (neuromancer/optimiser.py:135-165, 273-334)
# --------------------------------------------------------------------------
def adam_step(p, g, m, s, t, lr, mom, beta2, wd, apply_reg):
    """One Adam update exactly as optimiser.py:301-320.  t is the NEW step
    count (1 on the first call).  eps sits INSIDE the sqrt; L2 term is added
    outside the adaptive scaling and only where apply_reg."""
    factor = np.sqrt(1 - beta2 ** t) / (1 - mom ** t)
    new_m = mom * m + (1.0 - mom) * g
    new_s = beta2 * s + (1.0 - beta2) * g * g
    direction = factor * new_m / np.sqrt(new_s + EPS_ADAM)
    if apply_reg:
        mult = float(apply_reg) if apply_reg > 1 else 1.0
        new_p = p - lr * (direction + wd * p * mult)
    else:
        new_p = p - lr * direction
    return new_p, new_m, new_s


def sgd_step(p, g, d, lr, mom, wd, apply_reg):
    """optimiser.py:146-160: d' = g + mom*d ; p' = p - lr*(d' + wd*p)."""
    new_d = g + mom * d
    if apply_reg:
        mult = float(apply_reg) if apply_reg > 1 else 1.0
        new_p = p - lr * (new_d + wd * p * mult)
    else:
        new_p = p - lr * new_d
    return new_p, new_d


# --------------------------------------------------------------------------
# initialisation (variables.py:205-266 ; neural.py:146-204)
# --------------------------------------------------------------------------
def init_conv_params(rng, n_f, n_in, filter_shape, pool_shape, act='relu'):
    fov = float(np.prod(filter_shape))
    s = (n_in + float(n_f) / float(np.prod(pool_shape))) * fov
    w = rng.normal(0, np.sqrt(2.0 / s),
                   (n_f, n_in) + tuple(filter_shape)).astype(np.float32)
    if act == 'relu':
        b = np.full((n_f,), 1.0 / fov, np.float32)
    else:
        b = rng.uniform(-1e-6, 1e-6, (n_f,)).astype(np.float32)
    return w, b


# --------------------------------------------------------------------------
# whole sequential nets (examples/neuro3d_lite.py:48-60, neuro3d.py:48-63)
# --------------------------------------------------------------------------
NEURO3D_LITE = [  # (n_f, filter, pool, act)
    (20, (1, 4, 4), (1, 2, 2), 'relu'),
    (40, (3, 3, 3), (1, 2, 2), 'relu'),
    (150, (2, 4, 4), (2, 1, 1), 'relu'),
    (200, (1, 3, 3), (1, 1, 1), 'relu'),
    (200, (1, 3, 3), (1, 1, 1), 'relu'),
    (200, (1, 1, 1), (1, 1, 1), 'relu'),
    (2, (1, 1, 1), (1, 1, 1), 'lin'),
]
NEURO3D = [
    (20, (1, 6, 6), (1, 2, 2), 'relu'),
    (30, (1, 5, 5), (1, 2, 2), 'relu'),
    (40, (1, 5, 5), (1, 1, 1), 'relu'),
    (80, (4, 4, 4), (2, 1, 1), 'relu'),
    (100, (3, 4, 4), (1, 1, 1), 'relu'),
    (100, (3, 4, 4), (1, 1, 1), 'relu'),
    (150, (2, 4, 4), (1, 1, 1), 'relu'),
    (200, (1, 4, 4), (1, 1, 1), 'relu'),
    (200, (1, 4, 4), (1, 1, 1), 'relu'),
    (200, (1, 1, 1), (1, 1, 1), 'relu'),
    (2, (1, 1, 1), (1, 1, 1), 'lin'),
]


def net_out_shape(spec, in_spatial):
    """valid-conv + pool shape rule (neural.py:725-750); raises like the
    reference on indivisible pooling."""
    s = list(in_spatial)
    for n_f, k, p, _ in spec:
        for i in range(3):
            v = s[i] - k[i] + 1
            if v % p[i] != 0 or v <= 0:
                raise ValueError("Cannot pool axis %d of length %d by %d "
                                 "after kernel %d" % (i, s[i], p[i], k[i]))
            s[i] = v // p[i]
    return tuple(s)


def init_net(spec, n_in=1, seed=1):
    rng = np.random.RandomState(seed)
    params = []
    for n_f, k, p, act in spec:
        params.append(init_conv_params(rng, n_f, n_in, k, p, act))
        n_in = n_f
    return params


def net_fwd(spec, params, x, rnd=None, rnd_layers=None):
    """``rnd`` / ``rnd_layers``: operand rounding (conv_node_fwd) for the layers whose index is in
    ``rnd_layers`` (default: all)"""
    caches = []
    h = np.asarray(x, F64)
    for i, ((n_f, k, p, act), (w, b)) in enumerate(zip(spec, params)):
        ri = rnd if (rnd is not None and (rnd_layers is None or i in rnd_layers)) else None
        out, cache = conv_node_fwd(h, w, b, p, act, rnd=ri)
        caches.append((h, cache))
        h = out
    return h, caches


def net_loss_and_grads(spec, params, x, target, rnd=None, rnd_layers=None):
    """loss (softmax+NLL+aggregate) and gradients wrt every (w, b)."""
    logits, caches = net_fwd(spec, params, x, rnd=rnd, rnd_layers=rnd_layers)
    loss, dlogits, probs = nll_loss_and_grad(logits, target)
    grads = [None] * len(spec)
    d = dlogits
    for i in reversed(range(len(spec))):
        n_f, k, p, act = spec[i]
        w, b = params[i]
        h, cache = caches[i]
        ri = rnd if (rnd is not None and (rnd_layers is None or i in rnd_layers)) else None
        d, dw, db = conv_node_bwd(d, h, np.asarray(w, F64), b, cache, p, act,
                                  need_dx=(i > 0), rnd=ri)
        grads[i] = (dw, db)
    return loss, grads, probs


def net_train_steps(spec, params, x, target, n_steps, lr=5e-4, mom=0.9,
                    beta2=0.999, wd=0.5e-4):
    """n Adam steps on one fixed batch (examples/neuro3d.py:33-39 values).
    Parameters are kept in float64 between steps (oracle precision)."""
    P = [(np.asarray(w, F64), np.asarray(b, F64)) for w, b in params]
    M = [(np.zeros_like(w), np.zeros_like(b)) for w, b in P]
    S = [(np.zeros_like(w), np.zeros_like(b)) for w, b in P]
    losses = []
    for t in range(1, n_steps + 1):
        loss, grads, _ = net_loss_and_grads(spec, P, x, target)
        losses.append(loss)
        for i in range(len(P)):
            w, b = P[i]
            dw, db = grads[i]
            w, mw, sw = adam_step(w, dw, M[i][0], S[i][0], t, lr, mom, beta2,
                                  wd, True)
            b, mb, sb = adam_step(b, db, M[i][1], S[i][1], t, lr, mom, beta2,
                                  wd, False)
            P[i], M[i], S[i] = (w, b), (mw, mb), (sw, sb)
    return losses, P


# --------------------------------------------------------------------------
# dense prediction (node_basic.py:805-1012, without MFP)
# --------------------------------------------------------------------------
def net_geometry(spec):
    """(strides, fov, offsets) of a valid conv/pool stack, per spatial axis:
    fov grows by (k-1)*stride_so_far then by (p-1)*stride_so_far, strides multiply by
    the pool factors (neural.py:725-764); offsets = fov // 2 (graphutils.py:98-104)."""
    strides, fov = np.ones(3, np.int64), np.ones(3, np.int64)
    for n_f, k, p, _ in spec:
        fov = fov + (np.asarray(k) - 1) * strides
        fov = fov + (np.asarray(p) - 1) * strides
        strides = strides * np.asarray(p)
    return strides, fov, fov // 2


def predict_dense(spec, params, raw, patch_sp, as_uint8=False, pad_raw=False):
    """Restatement of Node.predict_dense / _predict_densetile for a sequential net
    (probabilities = softmax of net_fwd): block tiling with zero-padded end blocks,
    one forward pass per stride offset, outputs interleaved at [off::stride]."""
    raw = np.asarray(raw)
    m = 255.0 if raw.dtype.kind in 'iu' else 1.0
    raw = raw.astype(np.float32) / np.float32(m)
    strides, fov, offset = net_geometry(spec)
    if pad_raw:
        raw = np.pad(raw, [(0, 0)] + [(int(o), int(o)) for o in offset], mode='symmetric')
    ps = np.asarray(patch_sp, np.int64)
    out_sh = np.asarray(net_out_shape(spec, patch_sp), np.int64)
    n_lab = spec[-1][0]
    tile_sh = ps + strides - 1
    prob_sh = out_sh * strides
    raw_sh = np.asarray(raw.shape[1:], np.int64)
    pred_sh = raw_sh - 2 * offset
    pred = np.zeros((n_lab,) + tuple(pred_sh), np.uint8 if as_uint8 else np.float32)
    n_t = [int(np.ceil(float(pred_sh[i]) / prob_sh[i])) for i in range(3)]
    for zt in range(n_t[0]):
        for xt in range(n_t[1]):
            for yt in range(n_t[2]):
                lo = np.array([zt, xt, yt]) * prob_sh
                tile = raw[:, lo[0]:lo[0] + tile_sh[0], lo[1]:lo[1] + tile_sh[1],
                           lo[2]:lo[2] + tile_sh[2]]
                pad = tile_sh - np.asarray(tile.shape[1:])
                if np.any(pad > 0):                     # end block: zero padding on the far side
                    tile = np.pad(tile, [(0, 0)] + [(0, int(q)) for q in pad], mode='constant')
                prob = np.zeros((n_lab,) + tuple(prob_sh), np.float32)
                for oz in range(strides[0]):
                    for ox in range(strides[1]):
                        for oy in range(strides[2]):
                            cut = tile[None, :, oz:oz + ps[0], ox:ox + ps[1], oy:oy + ps[2]]
                            logits, _ = net_fwd(spec, params, cut)
                            prob[:, oz::strides[0], ox::strides[1], oy::strides[2]] = \
                                softmax(logits, 1)[0].astype(np.float32)
                keep = np.minimum(prob_sh, pred_sh - lo)
                prob = prob[:, :keep[0], :keep[1], :keep[2]]
                if as_uint8:
                    prob = prob * np.float32(255)
                pred[:, lo[0]:lo[0] + keep[0], lo[1]:lo[1] + keep[1], lo[2]:lo[2] + keep[2]] = prob
    return pred


def predict_voxel(spec, params, raw, pos):
    """independent check of predict_dense: the prediction at dense position ``pos``
    is the net evaluated on the field-of-view patch that starts there."""
    strides, fov, offset = net_geometry(spec)
    z, x, y = (int(v) for v in pos)
    cut = np.asarray(raw, np.float32)[None, :, z:z + fov[0], x:x + fov[1], y:y + fov[2]]
    logits, _ = net_fwd(spec, params, cut)
    assert logits.shape[2:] == (1, 1, 1), logits.shape
    return softmax(logits, 1)[0, :, 0, 0, 0]
