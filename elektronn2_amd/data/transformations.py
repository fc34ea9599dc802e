"""Geometric augmentation with the reference's API (elektronn2/data/transformations.py):
4x4 homogeneous matrix builders, the random flip / swap / rotation / warp matrices,
``warp_slice`` and ``get_warped_slice``.

Design for MI355X: ``img`` / ``target`` are torch tensors RESIDENT on the GPU (a numpy
array is uploaded, which is only sensible for tests); the matrix algebra and the
out-of-bounds test on the eight patch corners run on the host in NumPy exactly as in
the reference (transformations.py:388-405), the gather is one launch of
``e2_warp_slice`` per array (csrc/warp.hip) that computes the source coordinate of every
destination voxel on the fly instead of materialising a (z,x,y,3) coordinate array.
Not covered (outside the hot path): vector-valued targets (``target_vec_ix``), the
max-kernel interpolation of skeleton channels (``last_ch_max_interp``), 2-d patches.
"""
from __future__ import annotations

import itertools
from functools import reduce

import numpy as np
import torch

__all__ = ['identity', 'translate', 'scale', 'scale_inv', 'rotate_x', 'rotate_y', 'rotate_z',
           'chain_matrices', 'get_random_rotmat', 'get_random_flipmat', 'get_random_swapmat',
           'get_random_warpmat', 'make_dest_corners', 'WarpingOOBError', 'warp_slice',
           'get_warped_slice']


class WarpingOOBError(ValueError):
    """the warped patch would read outside the source volume (transformations.py:256)"""


# ---- matrices (transformations.py:110-236) ---------------------------------------------
def identity():
    return np.eye(4, dtype=np.float32)


def translate(dz, dy, dx):
    m = np.eye(4, dtype=np.float32)
    m[0, 3], m[1, 3], m[2, 3] = dz, dy, dx
    return m


def scale(mz, my, mx):
    return np.diag(np.asarray([mz, my, mx, 1.0], dtype=np.float32))


def scale_inv(mz, my, mx):
    return np.diag(np.asarray([1.0 / mz, 1.0 / my, 1.0 / mx, 1.0], dtype=np.float32))


def _rot(i, j, a, sign=1.0):
    m = np.eye(4, dtype=np.float32)
    c, s = np.cos(a), np.sin(a)
    m[i, i] = m[j, j] = c
    m[i, j], m[j, i] = -sign * s, sign * s
    return m


def rotate_z(a):
    """rotation in the (x, y) plane (rows 1, 2)"""
    return _rot(1, 2, a)


def rotate_y(a):
    """rotation in the (z, x) plane (rows 0, 1)"""
    return _rot(0, 1, a)


def rotate_x(a):
    """rotation in the (z, y) plane (rows 0, 2), opposite handedness"""
    return _rot(0, 2, a, sign=-1.0)


def chain_matrices(mat_list):
    return reduce(np.dot, mat_list, identity())


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
    flips[3] = 1                       # never the homogeneous coordinate
    if no_x_flip:
        flips[2] = 1
    return np.diag(flips.astype(np.float32))


def get_random_swapmat(lock_z=False, rng=None):
    rng = np.random.RandomState() if rng is None else rng
    perms = [p + (3,) for p in itertools.permutations((0, 1, 2)) if not lock_z or p[0] == 0]
    return np.eye(4, dtype=np.float32)[list(perms[rng.randint(0, len(perms))])]


def get_random_warpmat(lock_z=False, perspective=False, amount=1.0, rng=None):
    """identity + uniform perturbation of +-0.1*amount (perspective row scaled by 0.05
    and clipped to +-3e-3).  Like the reference (transformations.py:221) the numbers come
    from the GLOBAL numpy generator, ``rng`` is accepted and ignored."""
    a = 0.1 * amount
    perturb = np.random.uniform(-a, a, (4, 4))
    perturb[3, 3] = 0
    if lock_z:
        perturb[0] = 0
        perturb[:, 0] = 0
    if not perspective:
        perturb[3] = 0
    perturb[3, :3] = np.clip(perturb[3, :3] * 0.05, -3e-3, 3e-3)
    return np.eye(4, dtype=np.float32) + perturb


def make_dest_corners(sh):
    """homogeneous coordinates of the 8 corners of a destination array of shape sh"""
    hi = np.asarray(sh, np.float64) - 1
    c = np.array([[a * hi[0], b * hi[1], d * hi[2], 1.0]
                  for a, b, d in itertools.product((0, 1), repeat=3)])
    return c


# ---- warp_slice (transformations.py:337-492) ---------------------------------------------
def _ctx():
    from ..neuromancer.plan import get_ctx
    return get_ctx()


def _dev(a, ctx):
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            a = a.to(ctx.device)
        return a if a.dtype == torch.float32 else a.float()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(ctx.device)


def _corner_check(ps, M, M_inv, sh, what):
    corners = np.dot(M_inv, make_dest_corners(ps).T).T
    if np.any(M[3, :3] != 0):
        corners = corners / corners[:, 3][:, None]
    corners = corners[:, :3]
    lo = np.min(np.floor(corners), 0).astype(np.int64)
    hi = np.max(np.ceil(corners + 1), 0).astype(np.int64)      # + 1: linear interpolation
    if np.any(lo < 0) or np.any(hi >= np.asarray(sh)):
        raise WarpingOOBError(what)
    return corners


def warp_slice(img, ps, M, target=None, target_ps=None, target_vec_ix=None,
               target_discrete_ix=None, last_ch_max_interp=False, ksize=0.5, out=None,
               target_out=None):
    """Cut a warped patch of spatial size ``ps`` out of ``img`` (f, z, x, y) -- and the
    matching centred patch of size ``target_ps`` out of ``target`` -- by reading the
    source at ``M^-1 . dest`` (trilinear for images and non-discrete targets, nearest
    for the target channels in ``target_discrete_ix``; default: all).  Raises
    WarpingOOBError when the patch would leave the volume.  Returns device tensors
    ``(img_new, target_new)`` (``out`` / ``target_out``: optional preallocated dense
    destinations, e.g. the static input buffers of a training plan)."""
    if target_vec_ix is not None or last_ch_max_interp:
        raise NotImplementedError("vector targets / max-kernel interpolation are outside "
                                  "the HIP hot path")
    ctx = _ctx()
    ps = tuple(int(p) for p in ps)
    if len(ps) != 3:
        raise NotImplementedError("warp_slice: 3-d patches only")
    img = _dev(img, ctx)
    if img.dim() == 3:
        img = img[None]
    if img.dim() != 4:
        raise ValueError('img wrong dim/shape')
    n_f, sh = img.shape[0], tuple(img.shape[1:])
    M = np.asarray(M)
    M_inv = np.linalg.inv(M.astype(np.float64)).astype(np.float32)
    persp = bool(np.any(M[3, :3] != 0))
    _corner_check(ps, M, M_inv, sh, "Out of bounds")
    img_new = out if out is not None else torch.empty((n_f,) + ps, dtype=torch.float32,
                                                      device=ctx.device)
    with torch.cuda.stream(ctx.stream) if ctx.stream is not None else _null():
        ctx.warp_slice(img, M_inv, persp, 0, (0, 0, 0), (0.0, 0.0, 0.0), img_new)
    if target is None:
        return img_new, None
    target = _dev(target, ctx)
    target_ps = tuple(int(p) for p in target_ps)
    n_t, tsh = target.shape[0], tuple(target.shape[1:])
    off = np.subtract(sh, tsh)
    off_ps = np.subtract(ps, target_ps)
    if np.any(np.mod(off, 2)) or np.any(np.mod(off_ps, 2)):
        raise ValueError("targets must be centered w.r.t. images")
    off, off_ps = off // 2, off_ps // 2
    # bounds of the centred sub-block of destination coordinates, in target coordinates
    sub = make_dest_corners(target_ps)
    sub[:, :3] += off_ps
    c = np.dot(M_inv, sub.T).T
    if persp:
        c = c / c[:, 3][:, None]
    c = c[:, :3] - off
    lo_t = np.floor(c.min(0)).astype(np.int64)
    hi_t = np.ceil(c.max(0) + 1).astype(np.int64)
    if np.any(lo_t < 0) or np.any(hi_t >= np.asarray(tsh)):
        raise WarpingOOBError("Out of bounds for target")
    mask = 0
    for k in range(n_t):
        if target_discrete_ix is None or k in target_discrete_ix:
            mask |= 1 << k
    target_new = (target_out if target_out is not None else
                  torch.empty((n_t,) + target_ps, dtype=torch.float32, device=ctx.device))
    with torch.cuda.stream(ctx.stream) if ctx.stream is not None else _null():
        ctx.warp_slice(target, M_inv, persp, mask, tuple(int(v) for v in off_ps),
                       tuple(float(v) for v in off), target_new)
    return img_new, target_new


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def random_warp_matrix(img_sh, ps, aniso_factor=2, sample_aniso=True, warp_amount=1.0,
                       lock_z=True, no_x_flip=False, perspective=False, target_sh=None,
                       target_ps=None, rng=None):
    """the transformation ``get_warped_slice`` draws (transformations.py:583-626):
    random centre such that image and target patch fit, then
    ``M = T_dest . S_dest . R . W . F . S . S_src . T_src``; the generator is called in
    the reference's order (centre z, y, x; flips; swap; rotation; warp)."""
    rng = np.random.RandomState() if rng is None else rng
    ps = np.asarray(ps)
    img_sh = np.asarray(img_sh)
    dest_center = ps.astype(np.float64) / 2
    src_remainder = np.mod(ps, 2).astype(np.float64) / 2
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
    S = identity() if no_x_flip else get_random_swapmat(lock_z, rng)
    if np.isclose(warp_amount, 0):
        R = W = identity()
    else:
        R = get_random_rotmat(lock_z, warp_amount, rng)
        W = get_random_warpmat(lock_z, perspective, warp_amount, rng)
    T_src = translate(-z, -y, -x)
    S_src = scale(aniso_factor, 1, 1)
    S_dest = scale(1.0 / aniso_factor, 1, 1) if sample_aniso else identity()
    T_dest = translate(dest_center[0], dest_center[1], dest_center[2])
    return chain_matrices([T_dest, S_dest, R, W, F, S, S_src, T_src])


def get_warped_slice(img, ps, aniso_factor=2, sample_aniso=True, warp_amount=1.0,
                     lock_z=True, no_x_flip=False, perspective=False, target=None,
                     target_ps=None, target_vec_ix=None, target_discrete_ix=None, rng=None,
                     out=None, target_out=None):
    """random warp matrix + ``warp_slice`` (transformations.py:528-643)"""
    if len(ps) != 3:
        raise NotImplementedError("get_warped_slice: 3-d patches only")
    M = random_warp_matrix(tuple(img.shape[-3:]), ps, aniso_factor, sample_aniso, warp_amount,
                           lock_z, no_x_flip, perspective,
                           None if target is None else tuple(target.shape[-3:]), target_ps, rng)
    return warp_slice(img, ps, M, target=target, target_ps=target_ps,
                      target_vec_ix=target_vec_ix, target_discrete_ix=target_discrete_ix,
                      out=out, target_out=target_out)
