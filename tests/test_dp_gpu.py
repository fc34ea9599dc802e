"""GPU, world_size 2 on ONE card (gloo carries the exchange -- RCCL wants one GPU per
rank): the data-parallel training step of the launch plan.

  * replicas stay bit-identical over Adam steps,
  * the sliced exchange overlapped with the backward pass (plan._dp_cut, on by default)
    and the single all-reduce (plan option dp_overlap=False) give the same parameters,
  * both equal the torch-CPU port of the oracle stepping on the hand-averaged gradients
    of the two ranks' batches.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import e2_oracle as O
from oracle import torch_step as TS

pytestmark = pytest.mark.gpu
SP = (7, 47, 47)
STEPS = 3


def _data(rank):
    from elektronn2_amd import parallel
    rng = np.random.RandomState(parallel.rank_seed(3, rank))
    x = rng.rand(1, 1, *SP).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(O.NEURO3D_LITE, SP)).astype(np.float32)
    return x, t


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LOCAL_RANK="0")
    from elektronn2_amd import nets, neuromancer as nm, parallel
    assert parallel.init_from_env("gloo") == world
    x, t = _data(rank)
    out = {}
    for overlap in ("1", "0"):
        nm.set_plan_options(dp_overlap=overlap == "1")
        nm.model_manager.reset()
        m = nets.neuro3d_lite((None, 1) + SP, params=O.init_net(O.NEURO3D_LITE, 1, seed=5))
        m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
        m.loss(x, t)                              # builds the arena
        m.enable_data_parallel()
        losses = [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(STEPS)]
        plan = m.optimisers['Adam'].step.func
        out[overlap] = dict(P=m.P.cpu().numpy().copy(), losses=losses,
                            cut=plan._dp_cut(), n_graphs=len(plan._graphs or []))
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_overlapped_exchange():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=500) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for mode in ("1", "0"):
        assert np.array_equal(res[0][mode]["P"], res[1][mode]["P"]), "replicas diverged"
    # the overlapped path really was taken: a cut, and three graphs (fwd+bwd | bwd | update)
    k, lo = res[0]["1"]["cut"]
    assert k >= 1 and 0 < lo < res[0]["1"]["P"].size
    assert res[0]["1"]["n_graphs"] == 3 and res[0]["0"]["n_graphs"] == 2
    assert res[0]["0"]["cut"] is None
    a, b = res[0]["1"]["P"], res[0]["0"]["P"]
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()

    # oracle: two replicas stepping on the mean of their gradients
    torch.set_num_threads(8)
    nets_ = [TS.TorchNet(O.NEURO3D_LITE, O.init_net(O.NEURO3D_LITE, 1, seed=5))
             for _ in range(2)]
    data = [tuple(torch.tensor(v) for v in _data(r)) for r in range(2)]
    for _ in range(STEPS):
        for n, (x, t) in zip(nets_, data):
            n.loss_and_grads(x, t)
        for pa, pb in zip(nets_[0].w + nets_[0].b, nets_[1].w + nets_[1].b):
            g = (pa.grad + pb.grad) * 0.5
            pa.grad = g.clone(); pb.grad = g.clone()
        [n.adam() for n in nets_]
    want = [p.detach().numpy() for p in nets_[0].w + nets_[0].b]
    # arena layout: parameters in node order, w then b per node, 16-byte aligned slots
    nw = len(nets_[0].w)
    off = 0
    for i in range(nw):
        for ref in (want[i], want[nw + i]):
            got = a[off:off + ref.size].reshape(ref.shape)
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            assert err < 5e-4, (i, ref.shape, err)
            off += (ref.size + 3) // 4 * 4


def _ragged_data(rank):
    """rank 0: every voxel labelled; rank 1: three quarters unlabelled"""
    x, t = _data(rank)
    if rank == 1:
        rng = np.random.RandomState(99)
        t = t.copy()
        t[rng.rand(*t.shape) < 0.75] = -1
    return x, t


def _ragged_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LOCAL_RANK="0")
    from elektronn2_amd import nets, neuromancer as nm, parallel
    assert parallel.init_from_env("gloo") == world
    x, t = _ragged_data(rank)
    out = {}
    for fused in ("1", "0"):
        nm.set_plan_options(dp_fused_scale=fused == "1")
        nm.model_manager.reset()
        m = nets.neuro3d_lite((None, 1) + SP, params=O.init_net(O.NEURO3D_LITE, 1, seed=5))
        m.set_opt_meta_params('SGD', dict(lr=1e-2, mom=0.9, wd=0.0))
        m.loss(x, t)
        m.enable_data_parallel()
        losses = [float(m.trainingstep(x, t, optimiser='SGD')[0]) for _ in range(2)]
        plan = m.optimisers['SGD'].step.func
        out[fused] = dict(P=m.P.cpu().numpy().copy(), losses=losses, scale=plan._dp_scale(),
                          g_after=float(m.G.abs().max()))
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_with_ragged_label_counts():
    """The reference normalises the NLL by the labelled voxels of the WHOLE batch
    (loss.py:342-344).  Two ranks whose batches hold very different numbers of labelled voxels,
    two SGD steps (SGD: the scale of the gradient matters, Adam's first steps hide it): the
    plan's fused form -- unnormalised gradients, the count behind the arena, sum-all-reduce,
    division inside the optimiser kernel -- equals the scaling form of round 3 and the
    torch-CPU port stepping on the BATCH of the two samples; the arena is left zero."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=500) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert res[0]["1"]["scale"] == ('sum',) and res[0]["0"]["scale"] is None
    assert res[0]["1"]["g_after"] == 0.0                       # cleared by the optimiser launch
    for fused in ("1", "0"):
        assert np.array_equal(res[0][fused]["P"], res[1][fused]["P"]), "replicas diverged"
    a, b = res[0]["1"]["P"], res[0]["0"]["P"]
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
    # oracle: ONE net on the batch of both samples (nll_loss divides by the batch's labelled count)
    torch.set_num_threads(8)
    net = TS.TorchNet(O.NEURO3D_LITE, O.init_net(O.NEURO3D_LITE, 1, seed=5), dtype=torch.float64)
    xb = torch.tensor(np.concatenate([_ragged_data(r)[0] for r in range(2)]), dtype=torch.float64)
    tb = torch.tensor(np.concatenate([_ragged_data(r)[1] for r in range(2)]), dtype=torch.float64)
    params = net.w + net.b
    d = [torch.zeros_like(p_) for p_ in params]
    for _ in range(2):
        net.loss_and_grads(xb, tb)
        with torch.no_grad():
            for p_, d_ in zip(params, d):
                d_.mul_(0.9).add_(p_.grad)                     # optimiser.py:146-160, wd = 0
                p_.sub_(1e-2 * d_)
    want = [p_.detach().numpy() for p_ in params]
    nw = len(net.w)
    off = 0
    for i in range(nw):
        for ref in (want[i], want[nw + i]):
            got = a[off:off + ref.size].reshape(ref.shape)
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)
            # 5e-4 (the bound of the other step tests, test_model_gpu.TOL_ADAM), not 1e-4: the
            # second step's forward runs on parameters that carry the first step's f32 atomic
            # order, a relu unit within rounding of zero then falls on either side, and ONE such
            # flip moves a bias gradient -- a sum of ~1e5 signed terms -- by 1e-3 of its size
            # (DESIGN section 2).  Observed over five runs: 4e-5 ... 1.09e-4 on conv1's bias; what
            # the test distinguishes (count-weighted against plain mean, below) is 1e-2 apart.
            assert err < 5e-4, (i, ref.shape, err)
            off += (ref.size + 3) // 4 * 4
    # ... and it is NOT the plain mean of the two ranks' gradients (that is what ragged counts
    # distinguish): a net stepping on the hand-averaged per-rank gradients ends elsewhere
    nets_ = [TS.TorchNet(O.NEURO3D_LITE, O.init_net(O.NEURO3D_LITE, 1, seed=5), dtype=torch.float64)
             for _ in range(2)]
    for n_, r in zip(nets_, range(2)):
        x_, t_ = _ragged_data(r)
        n_.loss_and_grads(torch.tensor(x_, dtype=torch.float64), torch.tensor(t_, dtype=torch.float64))
    gm = 0.5 * (nets_[0].w[-1].grad + nets_[1].w[-1].grad).numpy()
    net2 = TS.TorchNet(O.NEURO3D_LITE, O.init_net(O.NEURO3D_LITE, 1, seed=5), dtype=torch.float64)
    net2.loss_and_grads(xb, tb)
    gb = net2.w[-1].grad.numpy()
    assert np.abs(gm - gb).max() > 1e-2 * np.abs(gb).max()


def _rccl_worker(port, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK="0")
    from elektronn2_amd import nets, neuromancer as nm
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    x, t = _data(0)
    t.flat[::7] = -1                              # some unlabelled voxels: the count rides along
    out = {}
    for mode in ("plain", "rccl-overlap", "rccl-single"):
        nm.set_plan_options(dp_overlap=mode != "rccl-single")
        nm.model_manager.reset()
        m = nets.neuro3d_lite((None, 1) + SP, params=O.init_net(O.NEURO3D_LITE, 1, seed=5))
        m.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
        m.loss(x, t)
        if mode != "plain":
            m.enable_data_parallel(exchange_at_world_1=True)
        losses = [float(m.trainingstep(x, t, optimiser='Adam')[0]) for _ in range(4)]
        # ... and two more through the several-steps entry point: one graph of two steps for the
        # plain plan; a data-parallel step (an exchange between its graphs) stays one step per launch
        l2, _ = m.trainingsteps(2, optimiser='Adam')
        losses += [float(v) for v in l2]
        plan = m.optimisers['Adam'].step.func
        out[mode] = dict(P=m.P.cpu().numpy().copy(), losses=losses, multi=sorted(plan._multi),
                         n_graphs=len(plan._graphs or []), backend=torch.distributed.get_backend())
    q.put(out)
    torch.distributed.destroy_process_group()


def test_exchange_over_rccl_with_a_one_rank_communicator():
    """RCCL itself, on the one GPU a test box has: a ONE-rank "nccl" group, the exchange forced
    (``enable_data_parallel(exchange_at_world_1=True)``).  The mean over one rank is the
    identity, so the segmented step -- graph A (forward + late backward) | async all-reduce
    of the arena's tail on RCCL's stream | graph B | all-reduce of the head + the labelled
    count | wait, scale | graph C (optimiser) -- must reproduce the plain step: same losses,
    same parameters, for the overlapped and the single-exchange form, eager call and graph
    replays.  What this covers that gloo cannot: RCCL initialisation, its stream ordering
    against the plan's stream between hipGraph launches, ``Work.wait()`` on device tensors,
    arena slices as collective buffers.  (What it cannot: more than one rank.)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    res = q.get(timeout=500)
    p.join(60)
    assert p.exitcode == 0
    assert res["rccl-overlap"]["backend"] == "nccl"
    assert res["plain"]["n_graphs"] == 1
    assert res["rccl-overlap"]["n_graphs"] == 3 and res["rccl-single"]["n_graphs"] == 2
    assert res["rccl-overlap"]["multi"] == [] and res["rccl-single"]["multi"] == [], res
    for mode in ("rccl-overlap", "rccl-single"):
        for a, b in zip(res[mode]["losses"], res["plain"]["losses"]):
            assert abs(a - b) <= 2e-5 * abs(b), (mode, res[mode]["losses"], res["plain"]["losses"])
        d = np.abs(res[mode]["P"] - res["plain"]["P"]).max()
        assert d <= 2e-5 * np.abs(res["plain"]["P"]).max(), (mode, d)
