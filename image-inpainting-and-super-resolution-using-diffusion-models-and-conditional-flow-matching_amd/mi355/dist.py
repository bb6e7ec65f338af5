"""Batch sharding across the GPUs of one node + the single collective of the sampling path.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the
tests).  Every image's trajectory is independent (GroupNorm and attention are per sample), so rank r takes the
contiguous slice [r*B/R, (r+1)*B/R) of the batch and runs the whole loop with no communication; the only
exchange is ONE all-gather of the finished (uint8 or fp32) shard.  The reference has
no distributed code at all.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* if world_size > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, rank: int | None = None, world_size: int | None = None) -> Tuple[int, int]:
    """Contiguous slice of `total` items owned by `rank` (sizes differ by at most one)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_batch(shard: torch.Tensor, total: int) -> torch.Tensor:
    """Collect per-rank shards (dim 0) into the full [total, ...] tensor on every rank: one all-gather.

    Equal shards use all_gather_into_tensor (a single RCCL call on one flat buffer); ragged shards are padded to
    the largest shard first (still one collective)."""
    r, w = world()
    if w == 1:
        return shard
    sizes = [shard_range(total, k, w) for k in range(w)]
    maxn = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == maxn for lo, hi in sizes):
        out = torch.empty((total,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(out, shard.contiguous())
        return out
    pad = torch.zeros((maxn,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    pad[: shard.shape[0]] = shard
    buf = torch.empty((w * maxn,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(buf, pad)
    return torch.cat([buf[k * maxn: k * maxn + (hi - lo)] for k, (lo, hi) in enumerate(sizes)], dim=0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])   # RCCL: pin the barrier's all-reduce to this rank's GPU
        else:
            dist.barrier()
