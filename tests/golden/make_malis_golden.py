"""Writes tests/golden/malis_reference.npz: the INPUT and EXPECTED-OUTPUT arrays of the
reference's own known-answer test for MALIS (/root/reference/tests/test_malis.py:36-77),
as data.  pos_true / g_true are the values that test asserts; the other arrays are its
inputs (segmentation ids, predicted affinities, neighbourhood)."""
import os

import numpy as np

nhood = np.array([[0, 1, 0], [0, 0, 1]], dtype=np.int32)
test_id2 = np.array([[[1, 1, 2, 2, 0, 3]] * 4], dtype=np.int32)
aff_pred = np.array([[[[1., 1., 1., 1., 0., 1.], [1., 1., 1., 1., 0., 1.],
                       [0.9, 0.8, 1., 1., 0., 1.], [0., 0., 0., 0., 0., 1.]]],
                     [[[1., 0., 1., 0.3, 0.4, 0.], [0.7, 0., 1., 0., 0., 0.],
                       [1., 0.2, 1., 0., 0., 0.], [1., 0., 1., 0., 0., 0.]]]], dtype=np.float32)
pos_true = np.array([[[[3, 2, 4, 0, 0, 3], [8, 0, 16, 0, 0, 2], [12, 0, 1, 0, 0, 1],
                       [0, 0, 0, 0, 0, 0]]],
                     [[[1, 0, 1, 0, 0, 0], [0, 0, 1, 0, 0, 0], [1, 0, 2, 0, 0, 0],
                       [1, 0, 3, 0, 0, 0]]]], dtype=np.int32)
g_true = pos_true.astype(np.float64)      # d(sum(pos * aff_pred))/d(aff_pred), counts are constants
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "malis_reference.npz"),
         nhood=nhood, seg_ids=test_id2, aff_pred=aff_pred, pos_true=pos_true, g_true=g_true)
