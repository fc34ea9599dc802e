"""BASELINE config 5's prediction half: dense prediction of a 512^3 volume with the
max-fragment-pooling rewrite of neuro3d_lite (synthetic volume, random weights).
usage: python tools/dense512.py [z x y] [--plain] [--mfma bf16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import elektronn2_amd
from elektronn2_amd import nets

args = [a for a in sys.argv[1:] if not a.startswith('--')]
shape = tuple(int(a) for a in args[:3]) if len(args) >= 3 else (512, 512, 512)
if '--mfma' in sys.argv:
    elektronn2_amd.set_mfma_dtype(sys.argv[sys.argv.index('--mfma') + 1])
np.random.seed(1)
if '--plain' in sys.argv:
    model = nets.neuro3d_lite((None, 1, 23, 183, 183))
else:
    model = nets.neuro3d_lite((1, 1, 24, 186, 186), mfp=True)
raw = np.random.RandomState(0).rand(1, *shape).astype(np.float32)
model.predict_dense(raw[:, :48, :224, :224])          # compile + tune
t0 = time.time()
pred = model.predict_dense(raw)
dt = time.time() - t0
assert pred.shape == (2,) + tuple(s - 2 * o for s, o in zip(shape, (2, 19, 19))), pred.shape
assert np.isfinite(pred).all() and abs(float(pred.sum(0).mean()) - 1.0) < 1e-4
print("dense prediction %s -> %s in %.2f s = %.1f M voxels/s (host volume in, host prediction out)"
      % (shape, pred.shape, dt, np.prod(pred.shape[1:]) / dt / 1e6))
