"""Target conversions of the data pipeline (reference: data/image.py)."""
from __future__ import annotations

import numpy as np

from .. import malis


def make_affinities(labels, nhood=None, size_thresh=1):
    """data/image.py:30-77: affinity graph + locally relabelled segmentation of a batch
    of ID volumes ``labels`` (bs, z, x, y).  ID 0 is background (never connected);
    edges leaving the volume are 0.  Returns ``aff`` (bs, #edges, z, x, y) int16 and
    ``seg`` (bs, z, x, y) int16 = connected components of ``aff`` (components smaller
    than ``size_thresh`` -> 0)."""
    labels = np.asarray(labels)
    if nhood is None:
        nhood = np.eye(3, dtype=np.int32)
    nhood = np.asarray(nhood, np.int32)
    aff = np.zeros((labels.shape[0], nhood.shape[0]) + labels.shape[1:], np.int16)
    seg = np.zeros(labels.shape, np.int16)
    for i, l in enumerate(labels):
        aff[i] = malis.seg_to_affgraph(l, nhood)
        seg[i], _ = malis.affgraph_to_seg(aff[i], nhood, size_thresh)
    return aff, seg
