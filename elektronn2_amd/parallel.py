"""Data-parallel glue (new: the reference is single-process).

One process per GPU; replicas of all weights; each rank draws its own
augmented sub-volume; ONE all-reduce (mean) of the flat gradient arena per step
(RCCL over xGMI on the GPU box -- backend "nccl" -- or gloo on CPU in tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def local_device(backend=None, local_rank=None, device_count=None):
    """Index of the GPU this rank drives: LOCAL_RANK.  A LOCAL_RANK the node has no GPU for
    is an ERROR under RCCL ("nccl": one GPU per rank -- a mis-bound 8-rank launch would
    otherwise put every rank on GPU 0 and only the rank list of the bench line would show it).
    Only a gloo rehearsal (backend "gloo" / E2_DIST_BACKEND=gloo: several ranks sharing the
    one card of a test box) wraps around."""
    idx = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    if device_count is None:
        if not torch.cuda.is_available():
            return idx
        device_count = torch.cuda.device_count()
    if idx < device_count:
        return idx
    if backend is None:
        backend = dist.get_backend() if dist.is_initialized() else \
            os.environ.get("E2_DIST_BACKEND", "nccl")
    if backend == "gloo" and device_count > 0:
        return idx % device_count
    raise RuntimeError("LOCAL_RANK=%d but this node shows %d GPU(s): launch one rank per visible "
                       "GPU (or set E2_DIST_BACKEND=gloo to rehearse several ranks on one card)"
                       % (idx, device_count))


def check_distinct_devices(ranks, backend):
    """``ranks``: one dict per rank with 'rank', 'host', 'pci' (and 'uuid').  Under RCCL two
    ranks on the same GPU of the same host are a mis-bound launch: raise, naming them.  (A
    gloo rehearsal shares the card on purpose.)"""
    if backend != "nccl":
        return
    seen = {}
    for r in ranks:
        key = (r.get("host"), r.get("pci") if r.get("pci") is not None else r.get("uuid"))
        if key[1] in (None, ""):
            continue
        if key in seen:
            raise RuntimeError("ranks %s and %s both run on GPU %s of host %s: one GPU per rank "
                               "is required under RCCL" % (seen[key], r.get("rank"), key[1], key[0]))
        seen[key] = r.get("rank")


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # (dmabuf IPC only on this pool's hosts)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device(backend))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return world


def allreduce_mean_(flat, group=None):
    """in-place mean over ranks of one flat tensor (the gradient arena)."""
    if not dist.is_initialized():
        return flat
    world = dist.get_world_size(group)
    if world == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat


class BucketedMean:
    """Mean over ranks of one flat tensor, exchanged as contiguous slices that are
    handed over as soon as they are final, so that the exchange of the gradients of the
    late layers runs under the backward pass of the early ones.

        bm = BucketedMean(G, group)
        ... backward of the late layers ...      bm.start(lo, n)   # G[lo:n] is final
        ... backward of the early layers ...     bm.start(0, lo)
        bm.finish()                              # waits, then scales by 1 / world

    start() enqueues an asynchronous all-reduce behind the work already queued on the
    current stream (RCCL runs it on its own stream); finish() makes the current stream
    wait for every slice."""

    def __init__(self, flat, group=None, count=None, spare=False, force=False, raw=False):
        """``count``: one-element tensor with this rank's number of LABELLED target voxels.
        The reference normalises the NLL by the labelled count of the WHOLE batch
        (loss.py:342-344); a rank's gradient is normalised by its own count, so with
        ragged counts the whole-batch gradient is the count-weighted mean
        sum_r n_r g_r / sum_r n_r, not the plain mean.  With ``count`` given,
        every slice is scaled by n_r before it is summed and the sum is divided by
        sum_r n_r afterwards; equal counts reduce to the plain mean.

        ``spare``: ``flat``'s storage holds at least one more float behind its last
        element (the model's gradient arena does).  The count then travels IN the first
        slice that reaches the arena's end -- no collective of its own, which at these
        sizes is pure latency; without a spare slot it takes a one-element all-reduce."""
        # raw: SUM only -- no pre-scaling by the count, no division afterwards.  The training
        # plan's form: its loss kernels leave the gradients unnormalised and put the labelled
        # count into the spare slot, the optimiser kernel divides by the summed count
        # (e2_set_loss_grad_mode / e2_adam_step_ex): no elementwise launch around the collective.
        self.raw = bool(raw)
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force: run the collectives in a one-rank group too (the sums are identities)
        self._skip = self.world == 1 and not (force and dist.is_initialized())
        self._work, self._covered = [], 0
        self._count, self._total = count, None
        self._ext = None
        if (count is not None or raw) and spare:
            n = flat.numel()
            self._ext = flat.as_strided((n + 1,), (1,), flat.storage_offset())

    def start(self, lo, hi):
        if self._skip or hi <= lo:
            return
        buf = self.flat
        if self.raw:
            if self._ext is not None and hi == self.flat.numel():     # the slot behind the tail
                buf, hi = self._ext, hi + 1
        elif self._count is not None:
            self.flat[lo:hi].mul_(self._count)               # n_r * g_r
            if self._ext is not None:
                if hi == self.flat.numel():                  # the count rides behind the tail
                    self._ext[hi:hi + 1].copy_(self._count)
                    buf, hi = self._ext, hi + 1
            elif self._total is None:
                self._total = self._count.detach().clone()
                dist.all_reduce(self._total, op=dist.ReduceOp.SUM, group=self.group)
        self._work.append(dist.all_reduce(buf[lo:hi], op=dist.ReduceOp.SUM,
                                          group=self.group, async_op=True))
        self._covered += min(hi, self.flat.numel()) - lo

    def finish(self):
        if self._skip:
            return self.flat
        for w in self._work:
            w.wait()
        if self._covered != self.flat.numel():
            raise RuntimeError("BucketedMean: slices cover %d of %d elements"
                               % (self._covered, self.flat.numel()))
        if self.raw:
            pass
        elif self._count is None:
            self.flat.mul_(1.0 / self.world)
        else:
            tot = self._ext[self.flat.numel():] if self._ext is not None else self._total
            self.flat.mul_(1.0 / torch.clamp(tot, min=1e-30))
        self._work, self._covered, self._total = [], 0, None
        return self.flat


def rank_seed(base_seed, rank=None):
    """independent data stream per rank (cnndata.py:193-204 reseeds per PID)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return int(base_seed) * 1000003 + int(rank) * 7919 + 17
