"""CPU, world_size 2, gloo: the data-parallel exchange step -- one all-reduce
(mean) of the flat gradient arena, then identical Adam updates on every rank
(SURVEY.md §8e).  The arithmetic of the step is the torch-CPU port of the oracle;
what is under test is elektronn2_amd.parallel and the replica invariants."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import e2_oracle as O
from oracle import torch_step as TS

SPEC = [(6, (1, 3, 3), (1, 2, 2), 'relu'), (8, (2, 3, 3), (1, 1, 1), 'relu'),
        (2, (1, 1, 1), (1, 1, 1), 'lin')]
SP = (4, 16, 16)


def _data(rank):
    from elektronn2_amd import parallel
    rng = np.random.RandomState(parallel.rank_seed(0, rank))
    x = rng.rand(1, 1, *SP).astype(np.float32)
    t = rng.randint(0, 2, (1, 1) + O.net_out_shape(SPEC, SP)).astype(np.float32)
    return torch.tensor(x), torch.tensor(t)


def _flat_grads(net):
    return torch.cat([p.grad.reshape(-1) for p in net.w + net.b])


def _set_flat_grads(net, flat):
    o = 0
    for p in net.w + net.b:
        n = p.numel()
        p.grad = flat[o:o + n].view_as(p).clone()
        o += n


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from elektronn2_amd import parallel
    assert parallel.init_from_env("gloo") == world
    net = TS.TorchNet(SPEC, O.init_net(SPEC, 1, seed=1))
    x, t = _data(rank)
    for _ in range(3):
        net.loss_and_grads(x, t)
        flat = _flat_grads(net)
        # the sliced exchange the launch plan uses (late layers first) gives the same mean
        sliced = flat.clone()
        bm = parallel.BucketedMean(sliced)
        lo = sliced.numel() // 3 // 4 * 4
        bm.start(lo, sliced.numel())
        bm.start(0, lo)
        bm.finish()
        parallel.allreduce_mean_(flat)
        assert torch.equal(sliced, flat)
        bm.start(lo, sliced.numel())
        try:
            bm.finish()
            raise AssertionError("partial coverage must raise")
        except RuntimeError as e:
            assert "cover" in str(e)
        _set_flat_grads(net, flat)
        net.adam()
    q.put((rank, torch.cat([p.detach().reshape(-1) for p in net.w + net.b]).numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_allreduce_mean_keeps_replicas_identical():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=240) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(res[0], res[1]), "replicas diverged"
    # single-process reference: average the two per-rank gradients by hand
    torch.set_num_threads(1)
    nets = [TS.TorchNet(SPEC, O.init_net(SPEC, 1, seed=1)) for _ in range(2)]
    data = [_data(r) for r in range(2)]
    for _ in range(3):
        fl = []
        for n, (x, t) in zip(nets, data):
            n.loss_and_grads(x, t)
            fl.append(_flat_grads(n))
        mean = (fl[0] + fl[1]) * 0.5
        for n in nets:
            _set_flat_grads(n, mean)
            n.adam()
    ref = torch.cat([p.detach().reshape(-1) for p in nets[0].w + nets[0].b]).numpy()
    assert np.abs(res[0] - ref).max() < 1e-6


def _ragged_data(rank):
    """rank r leaves a different share of its target unlabelled (-1)"""
    x, t = _data(rank)
    t = t.clone()
    t.view(-1)[::(2 + rank)] = -1
    return x, t


def _worker4(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from elektronn2_amd import parallel
    assert parallel.init_from_env("gloo") == world
    net = TS.TorchNet(SPEC, O.init_net(SPEC, 1, seed=1))
    x, t = _ragged_data(rank)
    net.loss_and_grads(x, t)
    flat = _flat_grads(net)
    count = (t >= 0).sum().to(torch.float32).view(1)
    # the launch plan's segment order (plan.py _segments, overlapped form): [forward +
    # backward of the late layers] -> exchange of the arena's tail starts -> [backward of
    # the early layers] -> exchange of the head -> wait, scale -> [optimiser]
    lo = flat.numel() // 4 // 4 * 4
    weighted = flat.clone()
    ex = parallel.BucketedMean(weighted, count=count)
    ex.start(lo, weighted.numel())
    ex.start(0, lo)
    ex.finish()
    plain = flat.clone()
    parallel.allreduce_mean_(plain)
    # the single-exchange form gives the same weighted result; so does the form in which
    # the count travels in the arena's spare slot instead of a collective of its own
    one = flat.clone()
    ex1 = parallel.BucketedMean(one, count=count)
    ex1.start(0, one.numel())
    ex1.finish()
    store = torch.full((flat.numel() + 4,), 7.0)
    rider = store[:flat.numel()]
    rider.copy_(flat)
    ex2 = parallel.BucketedMean(rider, count=count, spare=True)
    ex2.start(lo, rider.numel())
    ex2.start(0, lo)
    ex2.finish()
    assert (rider - weighted).abs().max() <= 1e-6 * weighted.abs().max()
    assert float(store[flat.numel()]) == float(sum((_ragged_data(r)[1] >= 0).sum() for r in range(world)))
    # (same numbers up to the summation order of gloo's ring over different slice sizes)
    assert (one - weighted).abs().max() <= 1e-6 * weighted.abs().max()
    q.put((rank, weighted.numpy(), plain.numpy(), float(count)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_four_ranks_ragged_labels_give_the_whole_batch_gradient():
    """world 4, every rank with a different number of labelled voxels: the reference
    normalises the NLL by the labelled count of the WHOLE batch (loss.py:342-344), so the
    data-parallel gradient must be the count-weighted mean of the per-rank gradients --
    checked against ONE process evaluating the batch of four; the plain mean (what equal
    counts reduce to) is measurably different here."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(4)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    res.sort(key=lambda r: r[0])
    assert len({r[3] for r in res}) == 4, "the counts must be ragged for this test"
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]), "ranks disagree on the exchanged gradient"
        assert np.array_equal(r[2], res[0][2])
    torch.set_num_threads(1)
    net = TS.TorchNet(SPEC, O.init_net(SPEC, 1, seed=1))
    xs, ts = zip(*[_ragged_data(r) for r in range(4)])
    net.loss_and_grads(torch.cat(xs), torch.cat(ts))        # whole-batch normalisation
    ref = _flat_grads(net).numpy()
    scale = np.abs(ref).max()
    assert np.abs(res[0][1] - ref).max() < 2e-6 * scale
    assert np.abs(res[0][2] - ref).max() > 1e-3 * scale      # the plain mean is NOT it


def test_rank_seeds_are_distinct_and_single_process_is_identity():
    from elektronn2_amd import parallel
    assert len({parallel.rank_seed(0, r) for r in range(8)}) == 8
    t = torch.arange(4.0)
    assert parallel.allreduce_mean_(t) is t and t.tolist() == [0, 1, 2, 3]
