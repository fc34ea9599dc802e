"""CPU ORACLE for the patch / warp augmentation row.  Test
infrastructure, NOT product code: only tests/ and bench.py's cpu_baseline leg import it.

NumPy restatement of the arithmetic of
  * data/transformations.py:337-492   warp_slice: inverse-mapped coordinates
    (float32, homogeneous divide for perspective matrices), bounds check on the
    destination corners, trilinear interpolation for images
    (map_coordinates_linear, :50-76: truncation, weights evaluated in float64 as
    numba promotes ``1 - du``), nearest neighbour for discrete targets
    (map_coordinates_nearest, :42-48: np.round = half-to-even), targets taken from
    the centred sub-block of the coordinates;
  * data/transformations.py:128-236   the matrix builders and the random
    flip / swap / rotation / warp matrices;
  * data/transformations.py:528-643   get_warped_slice: random centre, composition
    M = T_dest . S_dest . R . W . F . S . S_src . T_src, same RandomState call order;
  * data/cnndata.py:42-60             greyAugment.

PARITY STATUS: parity unpinned -- the reference module needs numba and (through
elektronn2/__init__) Theano, neither is installable here; its tests hold no vectors
for this path.  Pinned instead by exact identities (pure translations, axis
flips / swaps, integer scalings reproduce plain NumPy slicing / known interpolants),
see tests/test_warp.py.
"""
from __future__ import annotations

import itertools
from functools import reduce

import numpy as np

F32 = np.float32


class WarpingOOBError(ValueError):
    pass


# ---- matrix builders (transformations.py:110-236) ------------------------------------
def identity():
    return np.eye(4, dtype=F32)


def translate(dz, dy, dx):
    m = np.eye(4, dtype=F32)
    m[:3, 3] = (dz, dy, dx)
    return m


def scale(mz, my, mx):
    return np.diag(np.array([mz, my, mx, 1.0], dtype=F32))


def rotate_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=F32)


def rotate_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=F32)


def rotate_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=F32)


def chain_matrices(mats):
    return reduce(np.dot, mats, identity())


def get_random_rotmat(lock_z=False, amount=1.0, rng=None):
    rng = np.random.RandomState() if rng is None else rng
    gamma = rng.rand() * 2 * np.pi * amount
    if lock_z:
        return rotate_z(gamma)
    phi = rng.rand() * 2 * np.pi * amount
    theta = np.arcsin(rng.rand()) * amount
    return chain_matrices([rotate_z(gamma), rotate_y(-theta), rotate_z(-phi)])


def get_random_flipmat(no_x_flip=False, rng=None):
    rng = np.random.RandomState() if rng is None else rng
    flips = rng.binomial(1, 0.5, 4) * 2 - 1
    flips[3] = 1
    if no_x_flip:
        flips[2] = 1
    return np.diag(flips.astype(F32))


def get_random_swapmat(lock_z=False, rng=None):
    rng = np.random.RandomState() if rng is None else rng
    swaps = ([[0, 1, 2, 3], [0, 2, 1, 3]] if lock_z else
             [[0, 1, 2, 3], [0, 2, 1, 3], [1, 0, 2, 3], [1, 2, 0, 3], [2, 0, 1, 3],
              [2, 1, 0, 3]])
    return np.eye(4, dtype=F32)[swaps[rng.randint(0, len(swaps))]]


def get_random_warpmat(lock_z=False, perspective=False, amount=1.0, rng=None):
    """NB (reference quirk, transformations.py:221): the perturbation is drawn from the
    GLOBAL numpy generator, not from ``rng``."""
    amount *= 0.1
    perturb = np.random.uniform(-amount, amount, (4, 4))
    perturb[3, 3] = 0
    if lock_z:
        perturb[0] = 0
        perturb[:, 0] = 0
    if not perspective:
        perturb[3] = 0
    perturb[3, :3] *= 0.05
    np.clip(perturb[3, :3], -3e-3, 3e-3, out=perturb[3, :3])
    return np.eye(4, dtype=F32) + perturb


def make_dest_corners(sh):
    corners = np.array(list(itertools.product(*([0, 1],) * 3)), dtype=np.float64)
    corners = corners * (np.asarray(sh, np.float64) - 1)
    return np.hstack((corners, np.ones((8, 1))))


# ---- interpolation (transformations.py:42-76) -----------------------------------------
def _linear(src, coords, lo):
    """src (Z,X,Y) float32 cut that starts at ``lo``; coords (...,3) float32."""
    u = coords[..., 0] - lo[0]
    v = coords[..., 1] - lo[1]
    w = coords[..., 2] - lo[2]
    u0, v0, w0 = u.astype(np.int32), v.astype(np.int32), w.astype(np.int32)
    du = u.astype(np.float64) - u0
    dv = v.astype(np.float64) - v0
    dw = w.astype(np.float64) - w0
    u1, v1, w1 = u0 + 1, v0 + 1, w0 + 1
    s = src.astype(np.float64)
    val = (s[u0, v0, w0] * (1 - du) * (1 - dv) * (1 - dw) +
           s[u1, v0, w0] * du * (1 - dv) * (1 - dw) +
           s[u0, v1, w0] * (1 - du) * dv * (1 - dw) +
           s[u0, v0, w1] * (1 - du) * (1 - dv) * dw +
           s[u1, v0, w1] * du * (1 - dv) * dw +
           s[u0, v1, w1] * (1 - du) * dv * dw +
           s[u1, v1, w0] * du * dv * (1 - dw) +
           s[u1, v1, w1] * du * dv * dw)
    return val.astype(F32)


def _nearest(src, coords, lo):
    u = np.round(coords[..., 0] - lo[0]).astype(np.int32)
    v = np.round(coords[..., 1] - lo[1]).astype(np.int32)
    w = np.round(coords[..., 2] - lo[2]).astype(np.int32)
    return src[u, v, w].astype(F32)


def source_coords(ps, M):
    """(ps..., 3) float32 source coordinates of every destination voxel and M_inv."""
    M = np.asarray(M)
    M_inv = np.linalg.inv(M.astype(np.float64)).astype(F32)
    zz, xx, yy = np.mgrid[0:ps[0], 0:ps[1], 0:ps[2]]
    dest = np.stack([zz, xx, yy, np.ones_like(zz)], axis=-1).astype(F32)
    src = np.tensordot(dest, M_inv, axes=[[-1], [1]])
    if np.any(M[3, :3] != 0):
        src = src / src[..., 3][..., None]
    return src[..., :3].astype(F32), M_inv


def check_bounds(ps, M, sh):
    """the reference's corner test (transformations.py:396-405); returns (lo, hi)."""
    M = np.asarray(M)
    M_inv = np.linalg.inv(M.astype(np.float64)).astype(F32)
    corners = np.dot(M_inv, make_dest_corners(ps).T).T
    if np.any(M[3, :3] != 0):
        corners = corners / corners[:, 3][:, None]
    corners = corners[:, :3]
    lo = np.min(np.floor(corners), 0).astype(np.int64)
    hi = np.max(np.ceil(corners + 1), 0).astype(np.int64)
    if np.any(lo < 0) or np.any(hi >= np.asarray(sh)):
        raise WarpingOOBError("Out of bounds")
    return lo, hi


def warp_slice(img, ps, M, target=None, target_ps=None, target_discrete_ix=None):
    """(img_new (f,)+ps float32, target_new or None); raises WarpingOOBError."""
    ps = tuple(int(p) for p in ps)
    img = np.asarray(img)
    if img.ndim == 3:
        img = img[None]
    sh = img.shape[1:]
    lo, hi = check_bounds(ps, M, sh)
    coords, _ = source_coords(ps, M)
    cut = np.ascontiguousarray(img[:, lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1], F32)
    lo_f = lo.astype(F32)
    img_new = np.stack([_linear(cut[k], coords, lo_f) for k in range(img.shape[0])])
    if target is None:
        return img_new, None
    target = np.asarray(target)
    target_ps = tuple(int(p) for p in target_ps)
    off = np.subtract(sh, target.shape[1:])
    if np.any(np.mod(off, 2)):
        raise ValueError("targets must be centered w.r.t. images")
    off //= 2
    off_ps = np.subtract(ps, target_ps)
    if np.any(np.mod(off_ps, 2)):
        raise ValueError("targets must be centered w.r.t. images")
    off_ps //= 2
    ct = coords[off_ps[0]:off_ps[0] + target_ps[0], off_ps[1]:off_ps[1] + target_ps[1],
                off_ps[2]:off_ps[2] + target_ps[2]]
    lo_t = np.floor(ct.min(2).min(1).min(0) - off).astype(np.int64)
    hi_t = np.ceil(ct.max(2).max(1).max(0) - off + 1).astype(np.int64)
    if np.any(lo_t < 0) or np.any(hi_t >= np.asarray(target.shape[1:])):
        raise WarpingOOBError("Out of bounds for target")
    tcut = np.ascontiguousarray(target[:, lo_t[0]:hi_t[0] + 1, lo_t[1]:hi_t[1] + 1,
                                       lo_t[2]:hi_t[2] + 1], F32)
    ct = np.ascontiguousarray(ct, F32)
    lo_tf = (lo_t + off).astype(F32)
    n_t = target.shape[0]
    discrete = ([True] * n_t if target_discrete_ix is None
                else [i in target_discrete_ix for i in range(n_t)])
    target_new = np.stack([(_nearest if d else _linear)(tcut[k], ct, lo_tf)
                           for k, d in enumerate(discrete)])
    return img_new, target_new


def random_warp_matrix(img_sh, ps, aniso_factor=2, sample_aniso=True, warp_amount=1.0,
                       lock_z=True, no_x_flip=False, perspective=False, target_sh=None,
                       target_ps=None, rng=None):
    """the matrix get_warped_slice composes (transformations.py:583-626), with the same
    sequence of generator calls."""
    rng = np.random.RandomState() if rng is None else rng
    ps = np.asarray(ps)
    dest_center = ps.astype(np.float64) / 2
    src_remainder = np.mod(ps, 2).astype(np.float64) / 2
    img_sh = np.asarray(img_sh)
    if target_ps is not None:
        t_center = np.asarray(target_ps, np.float64) / 2
        off = np.subtract(img_sh, target_sh) // 2
        lo_pos = np.maximum(dest_center, t_center + off)
        hi_pos = np.minimum(img_sh - dest_center, np.asarray(target_sh) - t_center + off)
    else:
        lo_pos, hi_pos = dest_center, img_sh - dest_center
    z = rng.randint(lo_pos[0], hi_pos[0]) + src_remainder[0]
    y = rng.randint(lo_pos[1], hi_pos[1]) + src_remainder[1]
    x = rng.randint(lo_pos[2], hi_pos[2]) + src_remainder[2]
    F = get_random_flipmat(no_x_flip, rng)
    S = np.eye(4, dtype=F32) if no_x_flip else get_random_swapmat(lock_z, rng)
    if np.isclose(warp_amount, 0):
        R = W = np.eye(4, dtype=F32)
    else:
        R = get_random_rotmat(lock_z, warp_amount, rng)
        W = get_random_warpmat(lock_z, perspective, warp_amount, rng)
    T_src = translate(-z, -y, -x)
    S_src = scale(aniso_factor, 1, 1)
    S_dest = scale(1.0 / aniso_factor, 1, 1) if sample_aniso else identity()
    T_dest = translate(dest_center[0], dest_center[1], dest_center[2])
    return chain_matrices([T_dest, S_dest, R, W, F, S, S_src, T_src])


def grey_augment(d, channels, rng):
    """cnndata.py:42-60: d[ch] = clip(d[ch]*alpha + c, 0, 1) ** gamma."""
    if channels == []:
        return d
    k = len(channels)
    d = d.copy()
    alpha = 1 + (rng.rand(k) - 0.5) * 0.3
    c = (rng.rand(k) - 0.5) * 0.3
    gamma = 2.0 ** (rng.rand(k) * 2 - 1)
    d[channels] = d[channels] * alpha[:, None, None, None] + c[:, None, None, None]
    d[channels] = np.clip(d[channels], 0, 1)
    d[channels] = d[channels] ** gamma[:, None, None, None]
    return d
