"""How far do two IDENTICAL f32 runs of the training step drift apart?  (The weight gradients sum
with atomics, so two runs differ in summation order -- a few ulp per gradient -- and Adam's
1/sqrt(s + eps) and the relu decisions carry that forward.)  The tests that compare two plans of
the same model over several Adam steps (test_model_gpu / test_unet_config5_gpu / test_dp_gpu /
test_checkpoint) may hold that comparison no tighter than this spread; DESIGN finding 53 keeps
the numbers.

    python tools/aa_spread.py [pairs=4] [steps=9]
prints, per net, the largest relative loss difference at each step over the pairs and the largest
parameter / Adam-state difference (of the tensor's max) at the end."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from oracle import e2_oracle as O      # (initial parameters only: a tool, not the product)


def run(kind, steps, lr):
    from elektronn2_amd import nets, neuromancer as nm
    nm.model_manager.reset()
    rng = np.random.RandomState(5)
    if kind == 'unet':
        from test_unet_config5_gpu import small_unet
        sp, osp = (7, 30, 30), (3, 16, 16)
        np.random.seed(21)
        nm, inp, probs = small_unet(sp, n_out=2, batch=2)
        tgt = nm.Input((2, 1) + osp, 'b,f,z,x,y', name='target', dtype='int16')
        loss = nm.AggregateLoss(nm.MultinoulliNLL(probs, tgt, target_is_sparse=True))
        m = nm.model_manager.getmodel()
        m.designate_nodes(input_node=inp, target_node=tgt, loss_node=loss, prediction_node=probs)
        x = rng.rand(2, 1, *sp).astype(np.float32)
        t = rng.randint(0, 2, (2, 1) + osp).astype(np.int16)
    else:
        spec, sp = (O.NEURO3D_LITE, (7, 47, 47)) if kind == 'lite' else (O.NEURO3D, (17, 109, 109))
        params = O.init_net(spec, 1, seed=1)
        m = (nets.neuro3d_lite if kind == 'lite' else nets.neuro3d)((None, 1) + sp, params=params)
        x = rng.rand(1, 1, *sp).astype(np.float32)
        t = rng.randint(0, 2, (1, 1) + O.net_out_shape(spec, sp)).astype(np.float32)
    m.set_opt_meta_params('Adam', dict(lr=lr, mom=0.9, beta2=0.999, wd=0.5e-4))
    losses = [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(steps)]
    opt = m.optimisers['Adam']
    return (np.array(losses), [p.get_value() for p in m.loss_node.all_trainable_params.values()],
            opt.momentum.cpu().numpy().copy(), opt.squared_accum.cpu().numpy().copy())


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    for kind, lr in (('lite', 5e-4), ('full', 5e-4), ('unet', 2e-3)):
        dl = np.zeros(steps)
        dp = dm = ds = 0.0
        for _ in range(pairs):
            a, b = run(kind, steps, lr), run(kind, steps, lr)
            dl = np.maximum(dl, np.abs(a[0] - b[0]) / np.abs(b[0]))
            dp = max(dp, max(np.abs(u - v).max() / np.abs(v).max() for u, v in zip(a[1], b[1])))
            dm = max(dm, np.abs(a[2] - b[2]).max() / np.abs(b[2]).max())
            ds = max(ds, np.abs(a[3] - b[3]).max() / np.abs(b[3]).max())
        print("%-5s lr %g, %d pairs: loss rel diff per step %s | params %.2e | Adam m %.2e s %.2e"
              % (kind, lr, pairs, " ".join("%.1e" % v for v in dl), dp, dm, ds), flush=True)


if __name__ == "__main__":
    main()
