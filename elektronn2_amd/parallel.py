"""Data-parallel glue (new: the reference is single-process).

One process per GPU; replicas of all weights; each rank draws its own
augmented sub-volume; ONE all-reduce (mean) of the flat gradient arena per step
(RCCL over xGMI on the GPU box -- backend "nccl" -- or gloo on CPU in tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if torch.cuda.is_available():
        torch.cuda.set_device(min(int(os.environ.get("LOCAL_RANK", "0")),
                                  torch.cuda.device_count() - 1))
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

    def __init__(self, flat, group=None, count=None):
        """``count``: one-element tensor with this rank's number of LABELLED target voxels.
        The reference normalises the NLL by the labelled count of the WHOLE batch
        (loss.py:342-344); a rank's gradient is normalised by its own count, so with
        ragged counts the whole-batch gradient is the count-weighted mean
        sum_r n_r g_r / sum_r n_r, not the plain mean.  With ``count`` given, every slice
        is scaled by n_r * world / sum_r n_r before it is summed (one extra one-element
        all-reduce per step); equal counts reduce to the plain mean."""
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._work, self._covered = [], 0
        self._count, self._weight = count, None

    def _scale_slice(self, lo, hi):
        if self._count is None:
            return
        if self._weight is None:
            tot = self._count.detach().clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
            self._weight = self._count.detach() * float(self.world) / torch.clamp(tot, min=1e-30)
        self.flat[lo:hi].mul_(self._weight)

    def start(self, lo, hi):
        if self.world == 1 or hi <= lo:
            return
        self._scale_slice(lo, hi)
        self._work.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM,
                                          group=self.group, async_op=True))
        self._covered += hi - lo

    def finish(self):
        if self.world == 1:
            return self.flat
        for w in self._work:
            w.wait()
        if self._covered != self.flat.numel():
            raise RuntimeError("BucketedMean: slices cover %d of %d elements"
                               % (self._covered, self.flat.numel()))
        self._work, self._covered, self._weight = [], 0, None
        self.flat.mul_(1.0 / self.world)
        return self.flat


def rank_seed(base_seed, rank=None):
    """independent data stream per rank (cnndata.py:193-204 reseeds per PID)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return int(base_seed) * 1000003 + int(rank) * 7919 + 17
