"""BASELINE config 5's prediction half: dense prediction of a 512^3 volume (synthetic
volume, random weights) with
  * the examples/unet3d.py net, tiled (default): stride-1 UpConv net, plain tiles that overlap
    by input - output extent, built at a patch far larger than the training one (--patch);
  * --net lite: the max-fragment-pooling rewrite of neuro3d_lite (or --plain: stride offsets).
usage: python tools/dense512.py [z x y] [--net unet3d|lite] [--patch z,x,y] [--plain] [--mfma bf16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import elektronn2_amd
from elektronn2_amd import nets


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


skip = set()
for flag in ('--net', '--patch', '--mfma'):
    if flag in sys.argv:
        skip.add(sys.argv.index(flag) + 1)
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith('--') and i not in skip]
shape = tuple(int(a) for a in args[:3]) if len(args) >= 3 else (512, 512, 512)
if opt('--mfma'):
    elektronn2_amd.set_mfma_dtype(opt('--mfma'))
net = opt('--net', 'unet3d')
np.random.seed(1)
raw = np.random.RandomState(0).rand(1, *shape).astype(np.float32)
if net == 'unet3d':
    patch = tuple(int(v) for v in opt('--patch', '212,228,228').split(','))
    model = nets.unet3d((None, 1) + patch)
    offs = (44, 44, 44)
    osp = tuple(model.prediction_node.shape.spatial_shape)
    model.predict_dense(raw[:, :patch[0], :patch[1], :min(shape[2], patch[2] + osp[2])])   # compile + tune
else:
    if '--plain' in sys.argv:
        model = nets.neuro3d_lite((None, 1, 23, 183, 183))
    else:
        model = nets.neuro3d_lite((1, 1, 24, 186, 186), mfp=True)
    offs = (2, 19, 19)
    model.predict_dense(raw[:, :48, :224, :224])          # compile + tune
t0 = time.time()
pred = model.predict_dense(raw)
dt = time.time() - t0
assert pred.shape == (2,) + tuple(s - 2 * o for s, o in zip(shape, offs)), pred.shape
assert np.isfinite(pred).all() and abs(float(pred.sum(0).mean()) - 1.0) < 1e-4
print("dense prediction %s %s -> %s in %.2f s = %.1f M voxels/s (host volume in, host prediction out)"
      % (net, shape, pred.shape, dt, np.prod(pred.shape[1:]) / dt / 1e6))
