"""soak: N training steps of neuro3d_lite@183 on random patches of a synthetic volume
(device-resident sampler, warp + grey augmentation), checking that the loss stays finite
and falls and that the captured graphs keep replaying.
usage: soak.py [steps] [--mfma bf16] [--async] [--steps-per-launch k]
   --async: trainingstep(sync=False); --steps-per-launch k: Model.trainingsteps(k, sync=False) fed by a
   RingFeeder (the sampler fills the ring's next k slots while the launch of k steps runs)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as ndi
import elektronn2_amd
from elektronn2_amd import nets
from elektronn2_amd.data import PatchSampler, RingFeeder

steps = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 2000
if '--mfma' in sys.argv:
    elektronn2_amd.set_mfma_dtype(sys.argv[sys.argv.index('--mfma') + 1])
np.random.seed(0)
model = nets.neuro3d_lite((None, 1, 23, 183, 183))
model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
rng = np.random.RandomState(0)
vol = ndi.gaussian_filter(rng.rand(60, 400, 400).astype(np.float32), 3.0)
vol = ((vol - vol.min()) / (vol.max() - vol.min()))[None]
lab = (vol > np.median(vol)).astype(np.float32)
tn = model.target_node
smp = PatchSampler([vol], [lab], model.input_node.shape.spatial_shape, tn.shape.strides,
                   tn.shape.offsets, seed=1)
kpl = int(sys.argv[sys.argv.index('--steps-per-launch') + 1]) if '--steps-per-launch' in sys.argv else 1
t0 = time.time()
losses = []
if kpl > 1:
    d, t = smp.getbatch(1, 'train', grey_augment_channels=[0], warp=0.5)
    losses.append(float(model.trainingstep(d, t, optimiser='Adam')[0]))       # builds the plan
    feeder = RingFeeder(smp, model, 'Adam', kpl, grey_augment_channels=[0], warp=0.5)
    feeder.fill()
    t0, dev_t, n_dev = time.time(), 0.0, 0
    for b in range(steps // kpl):
        got, tl = model.trainingsteps(kpl, optimiser='Adam', ring=feeder.ring, sync=False)
        feeder.fill()
        if got is not None:
            losses += [float(v) for v in got]
            if tl:
                dev_t += tl; n_dev += kpl
        if (b + 1) % max(1, 500 // kpl) == 0:
            print("step %d  loss (mean of last 100) %.4f  %.3f ms/step wall, %.3f ms/step on the device" % (
                (b + 1) * kpl, np.mean(losses[-100:]), (time.time() - t0) / ((b + 1) * kpl) * 1e3,
                dev_t / max(n_dev, 1) * 1e3), flush=True)
    model.optimisers['Adam'].step.func.stream.synchronize()
else:
    for i in range(steps):
        d, t = smp.getbatch(1, 'train', grey_augment_channels=[0], warp=0.5)
        losses.append(float(model.trainingstep(d, t, optimiser='Adam', sync='--async' not in sys.argv)[0]))
        if (i + 1) % 500 == 0:
            print("step %d  loss (mean of last 100) %.4f  %.2f ms/step wall" % (
                i + 1, np.mean(losses[-100:]), (time.time() - t0) / (i + 1) * 1e3), flush=True)
P = model.P.cpu().numpy()
assert np.isfinite(losses).all() and np.isfinite(P).all()
assert np.mean(losses[-100:]) < 0.7 * np.mean(losses[:20]), (np.mean(losses[:20]), np.mean(losses[-100:]))
print("soak ok: %d steps, loss %.4f -> %.4f" % (steps, np.mean(losses[:20]), np.mean(losses[-100:])))
