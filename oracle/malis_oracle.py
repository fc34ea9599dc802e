"""CPU ORACLE for MALIS.  Test infrastructure, NOT product code.

Pure-Python restatement of malis/_malis_lib.cpp:38-125 (malis_loss_weights_cpp): edges
in DESCENDING weight order, Kruskal with union-find, every component carries a histogram
{ground-truth id: count}; the edge that joins two components is credited with the number
of voxel pairs across them whose ids are equal (pos) / different (neg).

PARITY STATUS: **pinned** by the reference's own known-answer test
(tests/test_malis.py:36-77: ``pos_true`` and ``g_true``), committed as data in
tests/golden/malis_reference.npz (generator: tests/golden/make_malis_golden.py).
Ties between equal weights are resolved by a STABLE sort on the edge index here; the
reference (and the product, csrc/malis.cpp) use std::sort, so the oracle is compared
with the product on distinct weights and on the golden vectors only.
"""
from __future__ import annotations

import numpy as np


def malis_loss_weights(seg, node1, node2, edge_weight, pos):
    n_vert = len(seg)
    parent = list(range(n_vert))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    hist = [({int(seg[i]): 1} if seg[i] != 0 else {}) for i in range(n_vert)]
    counts = np.zeros(len(edge_weight), np.uint64)
    valid = [e for e in range(len(edge_weight))
             if 0 <= node1[e] < n_vert and 0 <= node2[e] < n_vert]
    for e in sorted(valid, key=lambda e: -float(edge_weight[e])):
        a, b = find(int(node1[e])), find(int(node2[e]))
        if a == b:
            continue
        add = 0
        for ia, na in hist[a].items():
            for ib, nb in hist[b].items():
                if (ia == ib) if pos else (ia != ib):
                    add += na * nb
        counts[e] += add
        parent[b] = a
        for ib, nb in hist[b].items():
            hist[a][ib] = hist[a].get(ib, 0) + nb
        hist[b] = {}
    return counts


def maximin_pair_counts(seg, node1, node2, edge_weight, pos):
    """independent brute force for tiny graphs with DISTINCT weights: for every voxel pair
    find the maximin edge (the smallest edge on the best path = the edge that first
    connects them when edges are added in descending order) by re-running connectivity."""
    n_vert = len(seg)
    order = sorted([e for e in range(len(edge_weight))
                    if 0 <= node1[e] < n_vert and 0 <= node2[e] < n_vert],
                   key=lambda e: -float(edge_weight[e]))
    counts = np.zeros(len(edge_weight), np.uint64)
    comp = list(range(n_vert))
    for e in order:
        a, b = comp[int(node1[e])], comp[int(node2[e])]
        if a == b:
            continue
        for u in range(n_vert):
            if comp[u] != a:
                continue
            for v in range(n_vert):
                if comp[v] != b or seg[u] == 0 or seg[v] == 0:
                    continue
                if (seg[u] == seg[v]) == bool(pos):
                    counts[e] += 1
        comp = [a if c == b else c for c in comp]
    return counts


def edge_nodes(shape, nhood):
    """malis_utils.py:127-166: flat (node1, node2) of every (edge, voxel); node2 = -1
    where voxel + nhood[edge] leaves the volume"""
    nhood = np.asarray(nhood)
    nodes = np.arange(int(np.prod(shape))).reshape(shape)
    n1 = np.tile(nodes, (nhood.shape[0],) + (1,) * len(shape))
    n2 = np.full(n1.shape, -1, dtype=np.int64)
    for e, off in enumerate(nhood):
        src = tuple(slice(max(0, -int(o)), min(s, s - int(o))) for s, o in zip(shape, off))
        dst = tuple(slice(max(0, int(o)), min(s, s + int(o))) for s, o in zip(shape, off))
        n2[(e,) + src] = nodes[dst]
    return n1.ravel(), n2.ravel()


def malis_weights(aff_pred, aff_gt, seg_gt, nhood, unrestrict_neg=False):
    """malis_utils.py:377-459: pos pass on min(pred, gt), neg pass on pred (unrestricted)
    or max(pred, gt); (pos, neg) uint64 of aff_pred's shape"""
    sh = aff_pred.shape
    n1, n2 = edge_nodes(sh[1:], nhood)
    pred = np.asarray(aff_pred, np.float32).ravel()
    gt = np.asarray(aff_gt).ravel().astype(np.float32)
    seg = np.asarray(seg_gt).ravel()
    pos = malis_loss_weights(seg, n1, n2, np.minimum(pred, gt), 1)
    neg = malis_loss_weights(seg, n1, n2, pred if unrestrict_neg else np.maximum(pred, gt), 0)
    return pos.reshape(sh), neg.reshape(sh)


def malis_nll(probs, pos, neg, eps=1e-5):
    """loss.py:642-670 followed by AggregateLoss's mean (loss.py:1357-1363), float64:
    probs (2E, z, x, y) pair-softmax output (2e = disconnected, 2e+1 = affinity);
    returns (loss, dloss/dprobs).  nll.size cancels between the two nodes."""
    p = np.asarray(probs, np.float64)
    P, N = np.asarray(pos, np.float64), np.asarray(neg, np.float64)
    p0, p1 = p[0::2], p[1::2]
    n_tot = P.sum() + N.sum()
    with np.errstate(divide='ignore', invalid='ignore'):
        t = np.where(P != 0, P * np.log(p1 + eps), 0.0) + np.where(N != 0, N * np.log(p0 + eps), 0.0)
    loss = -t.sum() / (n_tot + eps)
    dp = np.zeros_like(p)
    dp[1::2] = -P / (p1 + eps) / (n_tot + eps)
    dp[0::2] = -N / (p0 + eps) / (n_tot + eps)
    return loss, dp
