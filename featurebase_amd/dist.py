"""Multi-GPU plumbing: one process per GPU, shards partitioned over ranks, partial counts
reduced with one collective (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests).

This mirrors the only cross-node exchange the reference's path has: `mapReduce` maps
per-shard functions on the node that owns each shard and folds count-valued results with
an associative `reduceFn` (executor.go:6449-6533, 5880); bitmap-valued results are never
exchanged (executor.go:1767).
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np


def rank_world() -> tuple:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shards_for_rank(n_shards: int, rank: int, world: int) -> List[int]:
    """Shard s lives on rank s % world (inside one node the reference's jump-hash placement,
    disco/hasher.go:15-24, is irrelevant; any fixed partition works because shards are
    independent)."""
    return list(range(rank, n_shards, world))


def init(backend: str, device=None):
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def all_reduce_counts(t):
    """In-place SUM of a count tensor over all ranks.  uint64 counts travel as int64 (the
    bit pattern of a wrap-around sum is the same)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_count_vector(local: np.ndarray, device: Optional[str] = None) -> np.ndarray:
    """Host convenience: all-reduce a numpy uint64 vector (GroupBy matrix, Sum triple …)."""
    import torch

    t = torch.from_numpy(np.ascontiguousarray(local).view(np.int64).copy())
    if device:
        t = t.to(device)
    all_reduce_counts(t)
    return t.cpu().numpy().view(np.uint64)
