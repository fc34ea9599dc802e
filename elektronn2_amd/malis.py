"""MALIS loss weights with the reference's Python API
(elektronn2/malis/malis_utils.py, malis/_malis.pyx): neighbourhood patterns, affinity
graphs from / to segmentations, and ``malis_weights``.  The sequential part (Kruskal with
union-find and per-component id histograms, connected components) is host C++ inside
libe2hip.so (csrc/malis.cpp) -- no GPU is involved; the array plumbing here is NumPy."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import backend

__all__ = ['mknhood2d', 'mknhood3d', 'nodelist_from_shape', 'seg_to_affgraph',
           'affgraph_to_seg', 'malis_loss_weights', 'connected_components', 'malis_weights']


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def mknhood2d(radius=1):
    """malis_utils.py:71-85: the first half (centre included, as in the reference) of
    the 2-d offsets within ``radius``, reversed"""
    r = int(np.ceil(radius))
    ax = np.arange(-r, r + 1)
    i, j = np.meshgrid(ax, ax)
    keep = (i ** 2 + j ** 2) <= radius ** 2
    i, j = i[keep].ravel(), j[keep].ravel()
    n = int(np.ceil(len(i) / 2.0))
    return np.ascontiguousarray(np.flipud(np.vstack((i[:n], j[:n])).T.astype(np.int32)))


def mknhood3d(radius=1):
    """malis_utils.py:87-109: the first half (centre INCLUDED: the reference computes
    ``ceil(len / 2)`` under true division) of the 3-d offsets within ``radius``,
    reversed; for radius 1: [[0,0,0],[-1,0,0],[0,-1,0],[0,0,-1]] (z, x, y)."""
    r = int(np.ceil(radius))
    ax = np.arange(-r, r + 1)
    i, j, k = np.meshgrid(ax, ax, ax)
    keep = (i ** 2 + j ** 2 + k ** 2) <= radius ** 2
    i, j, k = i[keep].ravel(), j[keep].ravel(), k[keep].ravel()
    n = int(np.ceil(len(i) / 2.0))
    nhood = np.vstack((k[:n], i[:n], j[:n])).T.astype(np.int32)
    return np.ascontiguousarray(np.flipud(nhood))


def _slices(shape, off):
    """(source, destination) slice tuples of an edge with displacement ``off``"""
    a = tuple(slice(max(0, -int(o)), min(s, s - int(o))) for s, o in zip(shape, off))
    b = tuple(slice(max(0, int(o)), min(s, s + int(o))) for s, o in zip(shape, off))
    return a, b


def nodelist_from_shape(shape, nhood):
    """malis_utils.py:127-166: start / end node index of every edge, -1 where the edge
    leaves the volume; arrays of shape (n_edge,) + shape, int32"""
    nhood = np.asarray(nhood)
    n_edge = nhood.shape[0]
    nodes = np.arange(int(np.prod(shape)), dtype=np.int32).reshape(shape)
    node1 = np.tile(nodes, (n_edge,) + (1,) * len(shape))
    node2 = np.full(node1.shape, -1, dtype=np.int32)
    for e in range(n_edge):
        a, b = _slices(shape, nhood[e])
        node2[(e,) + a] = nodes[b]
    return node1, node2


def seg_to_affgraph(seg_gt, nhood):
    """malis_utils.py:270-316: aff[e, v] = 1 iff v and v + nhood[e] carry the same
    non-zero id (edges leaving the volume: 0); int16"""
    nhood = np.ascontiguousarray(nhood, np.int32)
    seg_gt = np.asarray(seg_gt)
    shape = seg_gt.shape
    aff = np.zeros((nhood.shape[0],) + shape, dtype=np.int16)
    for e in range(nhood.shape[0]):
        a, b = _slices(shape, nhood[e])
        aff[(e,) + a] = (seg_gt[a] == seg_gt[b]) & (seg_gt[a] > 0) & (seg_gt[b] > 0)
    return aff


def malis_loss_weights(seg_true, node1, node2, edge_weight, pos):
    """_malis.pyx:42-66: uint64 impact counts per edge"""
    seg_true = np.ascontiguousarray(seg_true, np.int32)
    node1 = np.ascontiguousarray(node1, np.int32)
    node2 = np.ascontiguousarray(node2, np.int32)
    edge_weight = np.ascontiguousarray(edge_weight, np.float32)
    if not (node1.shape == node2.shape == edge_weight.shape and node1.ndim == 1):
        raise ValueError("malis_loss_weights: node lists and weights must be 1-d and equal")
    counts = np.zeros(edge_weight.shape[0], dtype=np.uint64)
    rc = backend.lib().e2_malis_loss_weights(int(seg_true.shape[0]), _ptr(seg_true),
                                             int(node1.shape[0]), _ptr(node1), _ptr(node2),
                                             _ptr(edge_weight), int(bool(pos)), _ptr(counts))
    if rc != 0:
        raise backend.E2Error("e2_malis_loss_weights failed (rc=%d)" % rc)
    return counts


def connected_components(n_vert, node1, node2, edge_weight, size_thresh=1):
    """_malis.pyx:70-94: (labels renumbered 0..k with 0 = background, sizes)"""
    node1 = np.ascontiguousarray(node1, np.int32)
    node2 = np.ascontiguousarray(node2, np.int32)
    edge_weight = np.ascontiguousarray(edge_weight, np.float32)
    seg = np.zeros(int(n_vert), dtype=np.int32)
    rc = backend.lib().e2_malis_connected_components(int(n_vert), int(node1.shape[0]),
                                                     _ptr(node1), _ptr(node2),
                                                     _ptr(edge_weight), int(size_thresh),
                                                     _ptr(seg))
    if rc != 0:
        raise backend.E2Error("e2_malis_connected_components failed (rc=%d)" % rc)
    unique, new_seg, sizes = np.unique(seg, return_inverse=True, return_counts=True)
    if 0 not in unique:                    # no background: label 0 must not be used
        new_seg[new_seg == 0] = unique[-1] + 1
    return new_seg.astype(np.int32), sizes


_edge_cache = {}


def _edges(vol_sh, nhood):
    key = (tuple(vol_sh), np.ascontiguousarray(nhood).tobytes())
    if key not in _edge_cache:
        n1, n2 = nodelist_from_shape(vol_sh, nhood)
        _edge_cache[key] = (n1.ravel(), n2.ravel())
    return _edge_cache[key]


def affgraph_to_seg(affinity_gt, nhood, size_thresh=1):
    """malis_utils.py:318-372: segmentation (connected components) of an affinity graph"""
    vol_sh = affinity_gt.shape[1:]
    node1, node2 = _edges(vol_sh, nhood)
    seg, sizes = connected_components(int(np.prod(vol_sh)), node1, node2,
                                      np.ascontiguousarray(affinity_gt, np.float32).ravel(),
                                      size_thresh)
    return seg.reshape(vol_sh), sizes


def malis_weights(affinity_pred, affinity_gt, seg_gt, nhood, unrestrict_neg=False):
    """malis_utils.py:377-459: (pos_counts, neg_counts), uint64, shape of affinity_pred.
    pos pass on min(pred, gt) (only true edges can carry must-link pairs), neg pass on
    max(pred, gt) -- or on pred itself when ``unrestrict_neg``."""
    sh = affinity_pred.shape
    if len(sh) != 4 or affinity_gt.shape != sh or tuple(seg_gt.shape) != tuple(sh[1:]):
        raise ValueError("malis_weights: affinity graphs (e, z, x, y) and seg (z, x, y) "
                         "must match")
    node1, node2 = _edges(sh[1:], nhood)
    gt = np.asarray(affinity_gt).ravel()
    pred = np.ascontiguousarray(affinity_pred, np.float32).ravel()
    seg = np.ascontiguousarray(seg_gt, np.int32).ravel()
    pos = malis_loss_weights(seg, node1, node2, np.minimum(pred, gt), 1)
    neg = malis_loss_weights(seg, node1, node2, pred if unrestrict_neg else np.maximum(pred, gt), 0)
    return pos.reshape(sh), neg.reshape(sh)
