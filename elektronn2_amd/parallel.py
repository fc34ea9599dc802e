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


def rank_seed(base_seed, rank=None):
    """independent data stream per rank (cnndata.py:193-204 reseeds per PID)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return int(base_seed) * 1000003 + int(rank) * 7919 + 17
