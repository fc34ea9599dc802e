#!/usr/bin/env python3
"""bench.py -- training throughput of the hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md §8d "C-lite@183"): the
neuro3d_lite net on synthetic (1,1,23,183,183) fp32 volumes -> (1,2,10,37,37),
one sample per GPU (weak scaling).  One "step" = forward + backward + (N>1: one
RCCL all-reduce of the flat gradient arena) + the reference's Adam update, all
inside the timed region, inputs already resident in HBM.

metric = training INPUT voxels per second (the reference's own definition,
training/trainer.py:283-284), whole job.  roofline: fp32-MFMA; achieved =
algorithmic fwd+bwd FLOPs per step (118.29 GF for this
workload) / mean step time measured with HIP events on the plan's stream.
cpu_baseline: the torch-CPU fp32 port of the same step (oracle/torch_step.py)
on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's intra-node
# transport fails with `hipIpcGetMemHandle: invalid argument` (it is exported on the pool's boxes;
# set here as well, before anything initialises HIP, so that a bare launcher inherits it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

WORKLOADS = {
    # name: (builder, input spatial, algorithmic fwd+bwd GFLOP/step  [SURVEY.md §8d])
    "lite183": ("neuro3d_lite", (23, 183, 183), 118.29),
    "full185": ("neuro3d", (23, 185, 185), 119.02),
    # BASELINE configs[2]; algorithmic GF with UpConv counted at its closed form
    "unet_lite140": ("unet3d_lite", (22, 140, 140), 397.8),
    # BASELINE configs[4]: examples/unet3d.py at its own input, UpConv p=(2,2,2)
    "unet132": ("unet3d", (116, 132, 132), 1576.9),
}
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak (spec; 155 measured)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak (--mfma bf16 only)
# roofline.traffic / roofline.mfma_util cannot be measured inside the timed run (PMC passes
# serialise the kernels): they are READ, at run time, from the rocprofv3 PMC summaries of this
# script that profiles/CURRENT.json names for the workload (tools/gpu_round.sh ->
# tools/adopt_profile.py), and only while the kernel sources and the shipped tilings are
# still the ones that were profiled (sources_sha16) -- otherwise null + "profile_stale".
#   traffic   = HBM-side bytes per step: 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md
#               "HBM") + WRITE_SIZE, separate --pmc passes, <tag>_bench_<workload>_pmc_traffic.csv
#   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x 2.4 GHz), summed over the
#               kernels of a step (<tag>_bench_<workload>_pmc_mfma.csv, tools/pmc_mfma.py): the
#               share of the chip's matrix-pipe cycles AT THE PEAK CLOCK in which an MFMA
#               executed.  (Round 2 divided by GRBM_GUI_ACTIVE / 8 instead, which reads 4-7 %
#               high on dispatches of 0.1-0.2 ms -- the guide says so, tools/clock_check.py
#               measured it: in-kernel clock 2.35-2.38 GHz, GRBM-derived 2.47-2.55.)


def sources_sha16():
    """identity of what a profile describes: kernel sources + shipped tilings"""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "elektronn2_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.hpp"))
                    + [os.path.join(ROOT, "elektronn2_amd", "tuned.json")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def recorded_profile(workload):
    """{"traffic", "mfma_util", "profile", "profile_stale"} from profiles/CURRENT.json"""
    import csv
    out = {"traffic": None, "mfma_util": None, "profile": None, "profile_stale": None}
    try:
        cur = json.load(open(os.path.join(ROOT, "profiles", "CURRENT.json"))).get(workload)
    except Exception:
        cur = None
    if not cur:
        return out
    base = os.path.join(ROOT, "profiles", "%s_bench_%s" % (cur["tag"], workload))
    out["profile"] = "profiles/%s_bench_%s_{pmc_traffic,pmc_mfma,kernel_stats}.csv" % (cur["tag"], workload)
    out["profile_stale"] = cur.get("sources_sha16") != sources_sha16()
    if out["profile_stale"]:
        return out
    try:
        tot = {}
        for r in csv.reader(open(base + "_pmc_traffic.csv")):
            if len(r) >= 6 and r[1] == "TOTAL":
                tot[r[0]] = float(r[5])                     # MiB per step
        out["traffic"] = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 * 1024
        for r in csv.reader(open(base + "_pmc_mfma.csv")):
            if r and r[0].startswith("TOTAL"):
                out["mfma_util"] = float(r[3])
    except Exception:
        pass
    return out


def algorithmic_gflop(model):
    """2*MAC per Conv node (``computational_cost`` = MACs, neural.py:767-778); wgrad = dgrad =
    fwd; no dgrad for the layer that reads the raw input."""
    tot = 0.0
    for n in model.nodes.values():
        if type(n).__name__ == 'Conv':
            tot += 2.0 * n.computational_cost * (2 if n.parent.is_source else 3)
    return tot / 1e9


def conv_params(model):
    """[(w, b)] of the Conv nodes in graph order (the initial weights, for the CPU leg)"""
    return [(n.w.get_value().copy(), n.b.get_value().copy()) for n in model.nodes.values()
            if type(n).__name__ == 'Conv']


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def host_cores():
    """CPU cores this process may actually use: cgroup quota if set, else the
    affinity mask, capped at the GPU box's per-GPU CPU share (16)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("E2_CPU_BASELINE_CORES", "16"))))


def dense_prediction_bench(args, rank, world):
    """SURVEY 8(f)-3 (not the headline metric): Model.predict_dense of neuro3d_lite at the
    BASELINE patch size over a (1,43,331,331) volume = 2x2x2 blocks x 32 stride offsets
    = 256 forward passes per prediction; one "step" = one whole-volume prediction."""
    from elektronn2_amd import nets
    sp = (23, 183, 183)
    mfp = args.workload == "dense183mfp"
    np.random.seed(1)
    if mfp:
        # the prediction-time rewrite with max-fragment pooling: the same (24,186,186) input
        # block, ONE pass with 32 fragments on the batch axis instead of 32 shifted passes
        model = nets.neuro3d_lite((1, 1, 24, 186, 186), mfp=True)
    else:
        model = nets.neuro3d_lite((None, 1) + sp)
    rng = np.random.RandomState(0)
    raw = rng.rand(1, 43, 331, 331).astype(np.float32)
    for _ in range(max(1, min(args.warmup, 2))):
        pred = model.predict_dense(raw)
    steps = max(1, min(args.steps, 5))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pred = model.predict_dense(raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    peak = PEAK_BF16_MFMA_TFLOPS if args.mfma == "bf16" else PEAK_FP32_MFMA_TFLOPS
    if mfp:
        fwd_gf = sum(2.0 * n.computational_cost for n in model.nodes.values()
                     if type(n).__name__ == 'Conv') / 1e9
        passes = 8
    else:
        fwd_gf = sum(2.0 * n.computational_cost for n in model.nodes.values()
                     if type(n).__name__ == 'Conv') / 1e9
        passes = 8 * 32
    out = {"metric": "dense_prediction_voxels_per_sec", "value": float(np.prod(pred.shape[1:])) / dt,
           "unit": "voxels/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
           "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
           "config": {"workload": "neuro3d_lite predict_dense (1,43,331,331)->(2,39,293,293), "
                                  + ("8 blocks, max-fragment pooling (32 fragments per pass), "
                                     if mfp else "8 blocks x 32 stride offsets, ")
                                  + "host volume in / host prediction out",
                      "algorithmic_gflop_per_pass": fwd_gf},
           "roofline": {"bound": "mfma", "achieved": passes * fwd_gf / dt / 1e3,
                        "peak": peak, "unit": "TFLOP/s",
                        "frac": passes * fwd_gf / dt / 1e3 / peak, "traffic": None}}
    if rank == 0:
        emit(out)


def dense_unet_bench(args, rank, world):
    """BASELINE configs[4]'s prediction half as stated: tiled dense prediction of a 512^3
    volume with the examples/unet3d.py net (not the headline metric).  A U-Net predicts at
    stride 1, so the tiles are plain blocks that overlap by input - output = 88 voxels per
    axis; the cost per predicted voxel falls with the tile size, and 288 GB of HBM hold
    the activations of tiles far larger than the (116,132,132) training patch: the net is
    built at (212,228,228) -> (124,140,140) (valid extents are 92 + 8 n), 4 x 4 x 4 tiles,
    several per launch.  One "step" = one whole-volume prediction, host volume in, host
    probabilities out."""
    from elektronn2_amd import nets
    patch = tuple(int(v) for v in os.environ.get("E2_DENSE_UNET_PATCH", "212,228,228").split(","))
    np.random.seed(1)
    model = nets.unet3d((None, 1) + patch)
    rng = np.random.RandomState(0)
    raw = rng.rand(1, 512, 512, 512).astype(np.float32)
    osp = tuple(model.prediction_node.shape.spatial_shape)
    model.predict_dense(raw[:, :patch[0], :patch[1], :patch[2] + osp[2]])   # compile (+ tune): 2 tiles
    steps = max(1, min(args.steps, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pred = model.predict_dense(raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert pred.shape == (2, 424, 424, 424), pred.shape
    assert np.isfinite(pred).all() and abs(float(pred.sum(0).mean()) - 1.0) < 1e-4
    # algorithmic forward FLOPs of one tile (SURVEY.md 8d convention): 2 * MAC per Conv; an
    # UpConv at its useful cost 2 * Cout * Cin * N_out (its computational_cost counts the
    # zero-stuffed dense form the CPU reference executes, prod(pool) times that)
    fwd_gf = sum(2.0 * n.computational_cost / (float(np.prod(n.pool_shape))
                                               if type(n).__name__ == 'UpConv' else 1.0)
                 for n in model.nodes.values() if type(n).__name__ in ('Conv', 'UpConv')) / 1e9
    n_tiles = int(np.prod([-(-424 // o) for o in osp]))
    peak = PEAK_BF16_MFMA_TFLOPS if args.mfma == "bf16" else PEAK_FP32_MFMA_TFLOPS
    out = {"metric": "dense_prediction_voxels_per_sec", "value": float(np.prod(pred.shape[1:])) / dt,
           "unit": "voxels/s", "n_gpus": 1, "steps": steps, "warmup": 1,
           "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.mfma, "data": "synthetic",
           "config": {"workload": "unet3d (examples/unet3d.py) predict_dense (1,512,512,512)->"
                                  "(2,424,424,424), %d tiles of %s -> %s, host volume in / host "
                                  "prediction out" % (n_tiles, patch, osp),
                      "algorithmic_gflop_per_tile": fwd_gf},
           "roofline": {"bound": "mfma", "achieved": n_tiles * fwd_gf / dt / 1e3,
                        "peak": peak, "unit": "TFLOP/s",
                        "frac": n_tiles * fwd_gf / dt / 1e3 / peak, "traffic": None}}
    if rank == 0:
        emit(out)


def warp_bench(args, rank, world):
    """SURVEY 8(f)-1 (not the headline metric): PatchSampler.getbatch at the BASELINE patch
    size -- random warp + perspective, image (23,183,183) trilinear + label target
    (19,145,145) nearest, grey augmentation -- from a (1,80,400,400) volume resident in HBM.
    One "step" = one patch.  roofline: HBM; algorithmic bytes = 8 B per voxel and channel
    (source read once + patch written) over the GPU time per patch (HIP events)."""
    from elektronn2_amd.data import PatchSampler
    from elektronn2_amd.neuromancer.plan import get_ctx
    rng = np.random.RandomState(0)
    vol = rng.rand(1, 80, 400, 400).astype(np.float32)
    lab = rng.randint(0, 2, (1, 80, 400, 400)).astype(np.float32)
    ps, strides, offsets = (23, 183, 183), (2, 4, 4), (2, 19, 19)
    np.random.seed(0)
    smp = PatchSampler([vol], [lab], ps, strides, offsets, seed=0)
    kw = dict(grey_augment_channels=[0], warp=True, warp_args={'sample_aniso': True,
                                                               'perspective': True})
    for _ in range(max(args.warmup, 3)):
        smp.getbatch(1, 'train', **kw)
    ctx = get_ctx()
    steps = max(args.steps, 20)
    torch.cuda.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(e0)
    for _ in range(steps):
        d, t = smp.getbatch(1, 'train', **kw)
    ctx.record(e1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    dev_ms = ctx.elapsed_ms(e0, e1) / steps
    vox = float(np.prod(ps)) + float(np.prod(smp.target_ps))
    out = {"metric": "augmented_patches_per_sec", "value": 1.0 / dt, "unit": "patches/s",
           "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": dt * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": "PatchSampler.getbatch patch (23,183,183) + target (19,145,145), "
                                  "warp + perspective + grey augment, volume (1,80,400,400) in HBM",
                      "input_voxels_per_sec": float(np.prod(ps)) / dt,
                      "oob_retries": smp.n_failed_warp},
           "roofline": {"bound": "hbm", "achieved": 8.0 * vox / (dev_ms * 1e-3) / 1e9, "peak": 8000.0,
                        "unit": "GB/s", "frac": 8.0 * vox / (dev_ms * 1e-3) / 1e9 / 8000.0,
                        "traffic": None, "device_ms_per_patch": dev_ms}}
    if not args.no_cpu_baseline:
        from oracle import warp_oracle as WO
        g = np.random.RandomState(1)
        ts = []
        while len(ts) < 3:
            M = WO.random_warp_matrix(vol.shape[1:], ps, 2, True, 1.0, True, False, True,
                                      lab.shape[1:], smp.target_ps, g)
            try:
                c0 = time.perf_counter()
                dd, tt = WO.warp_slice(vol, ps, M, target=lab, target_ps=smp.target_ps)
                WO.grey_augment(dd, [0], g)
                ts.append(time.perf_counter() - c0)
            except WO.WarpingOOBError:
                continue
        out["cpu_baseline"] = {"value": 1.0 / float(np.median(ts)), "unit": "patches/s", "cores": 1,
                               "kind": "port", "sample": "3 patches (median), NumPy restatement of "
                               "warp_slice + greyAugment (the reference uses numba-parallel gathers)"}
    if rank == 0:
        emit(out)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv, extra_env=None, timeout=3000):
    """`python bench.py --gpus N` without a launcher around it: start N fresh processes of
    this script (one per GPU: RANK = LOCAL_RANK = 0..N-1, WORLD_SIZE = N, rendezvous on
    127.0.0.1) and hand back (exit code, rank 0's stdout).  The parent never touches the
    GPU -- a process that has initialised HIP must not be replaced or forked -- and never
    retries: if any rank fails, the others are terminated and the code is non-zero."""
    import subprocess
    import tempfile
    port = _free_port()
    procs = []
    # rank 0's stdout goes to a FILE, read at the end: a pipe that nobody drains blocks rank 0
    # in write() once a library has printed ~64 KiB of banners there (NCCL_DEBUG), the other
    # ranks then hang in the collective and this loop would poll forever
    out_file = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n),
                    "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port)})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=env, stdout=out_file if r == 0 else subprocess.DEVNULL,
                                      text=True))
    out0, rc = "", 0
    t_end = None if timeout is None else time.time() + timeout
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                c = procs[r].poll()
                if c is None:
                    continue
                pending.discard(r)
                if c != 0 and rc == 0:
                    rc = c
            if rc != 0 or (t_end is not None and time.time() > t_end):
                rc = rc or 124
                break
            if pending:
                time.sleep(0.05)
        if rc == 0:
            out_file.seek(0)
            out0 = out_file.read()
    finally:
        out_file.close()
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        for pr in procs:
            try:
                pr.wait(timeout=20)
            except Exception:
                pr.kill()
    return rc, out0


def selftest_bench(args, rank, world):
    """`--workload selftest`: the multi-rank plumbing of this script WITHOUT a GPU (gloo on
    the CPU): rendezvous, barrier, MAX over ranks, the rank list in `config.ranks`.  Not a
    measurement; tests/test_bench_launcher.py runs it."""
    import torch.distributed as dist
    v = torch.tensor([float(rank)], dtype=torch.float64)
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()}
    ranks = [me]
    if world > 1:
        dist.barrier()
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    if os.environ.get("E2_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(3)
    if rank == 0 and os.environ.get("E2_SELFTEST_NOISE"):
        # a library that fills rank 0's stdout (NCCL_DEBUG banners) must not block the launch
        sys.stdout.write("x" * int(os.environ["E2_SELFTEST_NOISE"]) + "\n")
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
    if rank == 0:
        emit({"metric": "selftest", "value": float(v.item()), "unit": "rank",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"workload": "selftest (no GPU)", "ranks": ranks}})


def time_training(workload, args, rank, world, dev, steps, warmup):
    """Build the workload's net, capture its training step and time `steps` steps of it after
    `warmup` untimed ones (barrier + synchronize on both sides; HIP events on the plan's
    stream for the device time).  Inputs wait in HBM before the timed region starts."""
    from elektronn2_amd import parallel, nets
    from elektronn2_amd import neuromancer as nm
    builder, sp, gf_table = WORKLOADS[workload]
    nm.model_manager.reset()
    np.random.seed(1)                         # identical initial weights on every rank
    model = getattr(nets, builder)((None, 1) + sp)
    osp = tuple(model.prediction_node.shape.spatial_shape)
    if builder in ("unet3d_lite", "unet3d"):
        gflop = gf_table                      # UpConv / merge net: SURVEY.md §8d's figure
    else:
        gflop = algorithmic_gflop(model)
    params0 = conv_params(model)              # initial weights, for the CPU baseline leg
    model.set_opt_meta_params('Adam', dict(lr=5e-4, mom=0.9, beta2=0.999, wd=0.5e-4))
    opt = model.optimisers['Adam']
    opt.step.compile()
    plan = opt.step.func
    plan.use_graph = not args.no_graph
    rng = np.random.RandomState(parallel.rank_seed(0, rank))
    n_batches = 4
    xs = [torch.tensor(rng.rand(1, 1, *sp).astype(np.float32), device=dev)
          for _ in range(n_batches)]
    ts = [torch.tensor(rng.randint(0, 2, (1, 1) + osp).astype(np.float32), device=dev)
          for _ in range(n_batches)]
    plan.set_inputs([xs[0], ts[0]])          # builds plan + arena
    if world > 1:
        model.enable_data_parallel()
    elif args.exchange_at_1:
        model.enable_data_parallel(exchange_at_world_1=True)
    # the batches wait in HBM in the layout of the plan's input arena (image | target, each
    # slice 16-byte aligned): handing one over is ONE device copy into the static buffers
    arena = plan.input_arena
    staged = []
    for xb, tb in zip(xs, ts):
        flat = torch.zeros_like(arena)
        for node, src in zip(plan.inputs[:2], (xb, tb)):
            o, n_el = plan.input_slices[node]
            flat[o:o + n_el] = src.reshape(-1)
        staged.append(flat)

    # Steps per graph launch (--steps-per-launch, default 10; 1 = one launch per step as in rounds
    # 1-4).  With k > 1 on one GPU the staged batches are the slots of the plan's INPUT RING: every
    # captured step fetches its batch from HBM itself (e2_ring_fetch: the same bytes the copy below
    # moves), stores its loss in the device-side history (e2_hist_push), and k steps form one graph
    # -- the ~19 us the device idles between two launches are paid once per k steps (DESIGN finding
    # 55).  The data-parallel step (an exchange between its graphs) keeps one launch per segment.
    kpl = max(1, min(int(args.steps_per_launch), steps)) if (world == 1 and not args.exchange_at_1
                                                            and plan.use_graph) else 1
    done = [0]

    def one_step(i):
        with torch.cuda.stream(plan.stream):
            if plan._ring is None:
                arena.copy_(staged[i % n_batches], non_blocking=True)
            opt._ensure_state(plan)
            opt._sync_hyper(plan)
        plan.run()
        done[0] += 1

    def advance(n):
        """n steps: graphs of kpl steps, the remainder as single steps (which fetch from the ring too)"""
        if kpl == 1:
            for _ in range(n):
                one_step(done[0])
            return
        with torch.cuda.stream(plan.stream):
            opt._ensure_state(plan)
            opt._sync_hyper(plan)
        for _ in range(n // kpl):
            plan.run_steps(kpl)
            done[0] += kpl
        for _ in range(n % kpl):
            one_step(done[0])

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    if kpl > 1:
        # set-up, untimed and not counted as warm-up: the eager step, the capture of the single
        # step (with the ring attached: it contains the fetch), the capture of the kpl-step graph
        one_step(0)
        plan.set_input_ring(torch.stack(staged))
        plan.keep_loss_history()
        one_step(1)
        plan.run_steps(kpl)
        assert kpl in plan._multi, "the %d-step graph was not captured" % kpl
    advance(warmup)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ctx = plan.ctx
    e0, e1 = ctx.event(), ctx.event()
    old = ctx.stream
    t0 = time.perf_counter()
    ctx.set_stream(plan.stream); ctx.record(e0); ctx.set_stream(old)
    advance(steps)
    ctx.set_stream(plan.stream); ctx.record(e1); ctx.set_stream(old)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev_ms = ctx.elapsed_ms(e0, e1) / steps
    loss = float(plan.scratch[model.loss_node.parent[0], 'loss'].item())
    assert np.isfinite(loss), "non-finite loss"
    if kpl > 1:
        # every timed step took a batch and left a loss: the ring and the history counted them
        hist = plan.loss_history(min(steps, 256))
        assert np.isfinite(hist).all() and abs(float(hist[-1]) - loss) <= 1e-6 * abs(loss), (hist[-3:], loss)
    return dict(builder=builder, sp=sp, osp=osp, gflop=gflop, params0=params0, model=model,
                plan=plan, xs=xs, ts=ts, dt=dt, dev_ms=dev_ms, loss=loss, kpl=kpl)


_REAL_STDOUT = None


def claim_stdout():
    """stdout carries the ONE JSON line and nothing else: libraries print there too (RCCL's
    version banner at communicator creation, gloo's connection notes) -- everything written
    to fd 1 from here on goes to stderr, the line itself to the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="lite183", choices=sorted(WORKLOADS) + ["dense183", "dense183mfp", "dense512unet", "warp183", "selftest"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--steps-per-launch", type=int, default=int(os.environ.get("E2_STEPS_PER_LAUNCH", "10")),
                    help="training steps per graph launch at N = 1 (Plan.run_steps: batches from a "
                         "device-side ring, losses into a device-side history); 1 = one launch per step")
    ap.add_argument("--no-also", action="store_true",
                    help="lite183 at N = 1 also times neuro3d@185 and the two U-Nets "
                         "(roofline.also.<workload>); skip them")
    ap.add_argument("--also", default="full185,unet_lite140,unet132",
                    help="the workloads that ride along in the default N = 1 line")
    ap.add_argument("--exchange-at-1", action="store_true",
                    help="N = 1 only, not the headline: run the data-parallel form of the step "
                         "(segmented graphs + the RCCL all-reduces) over a ONE-rank nccl group -- "
                         "what the exchange machinery itself costs per step")
    ap.add_argument("--sources-sha", action="store_true",
                    help="print the identity of the kernel sources + shipped tilings and exit")
    ap.add_argument("--mfma", default=os.environ.get("E2_MFMA_DTYPE", "f32"), choices=["f32", "bf16"],
                    help="arithmetic of the conv GEMMs: f32 (default, the BASELINE metric) or "
                         "bf16 operands with f32 sums (SURVEY.md 8f-3; a separate, looser-"
                         "tolerance path -- never the headline number)")
    args = ap.parse_args()
    if args.sources_sha:
        print(sources_sha16())
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (nothing has touched the GPU yet)
        rc, line = launch_ranks(args.gpus, sys.argv[1:])
        if rc == 0:
            # rank 0's ONE JSON line goes to stdout; anything else a library printed there
            # (gloo's "[Gloo] Rank 0 is connected ..." banner) goes to stderr
            for l in line.splitlines():
                (sys.stdout if l.lstrip().startswith("{") else sys.stderr).write(l + "\n")
            sys.stdout.flush()
        sys.exit(rc)

    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks, or none: "
                 "`python bench.py --gpus N` starts its own)" % (args.gpus, world))
    if args.workload == "selftest":
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return selftest_bench(args, rank, world)

    from elektronn2_amd import parallel, nets
    from elektronn2_amd import neuromancer as nm

    if world > 1:
        # RCCL ("nccl") on the GPU node; E2_DIST_BACKEND=gloo only to rehearse the
        # multi-rank flow where the ranks have to share one GPU
        parallel.init_from_env(os.environ.get("E2_DIST_BACKEND", "nccl"))
    elif args.exchange_at_1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        torch.cuda.set_device(0)
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)

    bf16 = args.mfma == "bf16"
    if bf16:
        import elektronn2_amd
        elektronn2_amd.set_mfma_dtype("bf16")
    if args.workload in ("dense183", "dense183mfp"):
        return dense_prediction_bench(args, rank, world)
    if args.workload == "dense512unet":
        return dense_unet_bench(args, rank, world)
    if args.workload == "warp183":
        return warp_bench(args, rank, world)
    backend_name = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    dev = torch.device("cuda", parallel.local_device(backend_name))
    r = time_training(args.workload, args, rank, world, dev, args.steps, args.warmup)
    builder, sp, osp, gflop = r["builder"], r["sp"], r["osp"], r["gflop"]
    plan, dt, dev_ms, loss = r["plan"], r["dt"], r["dev_ms"], r["loss"]

    ranks = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        # who took part: one line per rank (device, PCI bus id, backend) gathered on rank 0,
        # so that a scaling run shows N distinct GPUs behind the N ranks
        import socket
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
              "device": dev.index, "name": props.name, "host": socket.gethostname(),
              "pci": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
              "backend": torch.distributed.get_backend(), "ms_per_step": dev_ms}
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, me)
        ranks = gathered
        # two ranks on one GPU under RCCL = a mis-bound launch: no line, non-zero exit
        parallel.check_distinct_devices(ranks, torch.distributed.get_backend())

    if rank != 0:
        return
    vox_per_step = float(np.prod((1, 1) + sp)) * world
    value = vox_per_step * args.steps / dt
    achieved = gflop / (dev_ms * 1e-3) / 1e3            # TFLOP/s per GPU
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    prof = recorded_profile(args.workload)
    out = {
        "metric": "training_input_voxels_per_sec",
        "value": value,
        "unit": "voxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16" if bf16 else "f32",
        "data": "synthetic",
        "config": {"workload": "%s (1,1,%d,%d,%d)->(1,2,%d,%d,%d) fwd+bwd+Adam, 1 sample/GPU%s"
                               % ((builder,) + tuple(sp) + tuple(osp) +
                                  ((", conv GEMM operands rounded to bf16, f32 sums, f32 tensors",)
                                   if bf16 else ("",))),
                   "parallelism": "dp%d" % world + (" (exchange forced over a one-rank RCCL group)"
                                                    if args.exchange_at_1 and world == 1 else ""),
                   "ranks": ranks,
                   "output_voxels_per_sec": float(np.prod(osp)) * world * args.steps / dt,
                   "hipgraph": bool(plan.use_graph), "steps_per_launch": r["kpl"], "final_loss": loss},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": None if bf16 else prof["traffic"],
                     "traffic_unit": "B/step: 2*FETCH_SIZE + WRITE_SIZE of the recorded rocprofv3 "
                                     "PMC passes (not measured in this run)",
                     "mfma_util": None if bf16 else prof["mfma_util"],
                     "mfma_util_unit": "MFMA-busy cycles / (1024 SIMDs x kernel time x 2.4 GHz), "
                                       "recorded PMC pass (not measured in this run)",
                     "profile": prof["profile"], "profile_stale": prof["profile_stale"],
                     "kernel": "training step (hipGraph): conv3d igemm fwd/dgrad/wgrad on "
                               + ("v_mfma_f32_16x16x16_bf16 / 16x16x32_bf16" if bf16 else
                                  "v_mfma_f32_16x16x4_f32") + " + pointwise + Adam",
                     "algorithmic_gflop_per_step": gflop,
                     "device_ms_per_step": dev_ms},
    }
    # what the CPU leg needs, on the host -- then the headline plan's graphs, arena and staged
    # batches are released before any further net is built
    cpu_in = (r["xs"][0].cpu().numpy(), r["ts"][0].cpu().numpy(), r["params0"])
    if (world == 1 and args.workload == "lite183" and not args.no_also and not bf16
            and not args.exchange_at_1):
        # The net the target NAMES (north_star: ">= 50 % MFMA roofline on the neuro3d 3D-conv
        # fwd+bwd", examples/neuro3d.py:51-63) and BASELINE configs 3 and 5 (examples/
        # unet3d_lite.py:59-98, examples/unet3d.py:61-100) ride in the same driver-run line: same
        # process, same protocol, each its own captured step and its own profile of record; the
        # headline fields above stay BASELINE configs[1].  A
        # failure in one of these legs is recorded under its name and never costs the headline.
        import gc
        for k in ("plan", "model", "xs", "ts"):
            r.pop(k, None)
        del plan
        out["roofline"]["also"] = {}
        for wl in args.also.split(","):
            gc.collect()
            torch.cuda.empty_cache()
            try:
                r2 = time_training(wl, args, rank, world, dev, args.steps, args.warmup)
                prof2 = recorded_profile(wl)
                ach2 = r2["gflop"] / (r2["dev_ms"] * 1e-3) / 1e3
                out["roofline"]["also"][wl] = {
                    "workload": "%s (1,1,%d,%d,%d)->(1,2,%d,%d,%d) fwd+bwd+Adam, 1 sample/GPU"
                                % ((r2["builder"],) + tuple(r2["sp"]) + tuple(r2["osp"])),
                    "steps": args.steps, "warmup": args.warmup, "steps_per_launch": r2["kpl"],
                    "ms_per_step": r2["dt"] / args.steps * 1e3, "device_ms_per_step": r2["dev_ms"],
                    "value": float(np.prod((1, 1) + r2["sp"])) * args.steps / r2["dt"], "unit": "voxels/s",
                    "algorithmic_gflop_per_step": r2["gflop"], "achieved": ach2, "peak": peak,
                    "frac": ach2 / peak, "mfma_util": prof2["mfma_util"], "traffic": prof2["traffic"],
                    "profile": prof2["profile"], "profile_stale": prof2["profile_stale"],
                    "final_loss": r2["loss"]}
                del r2
            except Exception as e:          # noqa: BLE001 -- recorded, the headline line survives
                out["roofline"]["also"][wl] = {"error": "%s: %s" % (type(e).__name__, e)}
    if builder in ("unet3d_lite", "unet3d"):
        args.no_cpu_baseline = True           # the CPU leg is the sequential-net port
    if world == 1 and not args.no_cpu_baseline:
        x, t, params0 = cpu_in
        # the ONLY use of oracle/ in this script: the CPU port, timed as the baseline
        from oracle import e2_oracle as O
        from oracle import torch_step as TS
        spec = O.NEURO3D_LITE if builder == "neuro3d_lite" else O.NEURO3D
        cores = host_cores()
        # BASELINE.md §3 protocol: same tensors, batch 1, 2 warm-up steps, median of 5
        med, all_t = TS.time_cpu_step(spec, params0, x, t, cores, warmup=2, steps=5)
        out["cpu_baseline"] = {
            "value": float(np.prod((1, 1) + sp)) / med, "unit": "voxels/s", "cores": cores,
            "kind": "port",
            "sample": "5 full training steps (median, after 2 warm-up steps) of the same "
                      "workload, torch-CPU fp32 (oneDNN) port of the reference step; Theano "
                      "itself is not installable offline (BASELINE.md §3), oneDNN direct "
                      "convolution is expected to be faster than Theano's conv3d2d",
            "s_per_step": med,
            "gflops": gflop / med,
            "cpu_model": cpu_model(),
            "torch": torch.__version__,
        }
    emit(out)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
