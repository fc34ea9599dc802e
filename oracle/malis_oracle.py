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
