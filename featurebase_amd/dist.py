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


class BucketedCountReducer:
    """Cross-GPU sum of per-step partial counts with ONE collective per `bucket` steps.

    The only exchange of the path is the reduce of count-valued partial results
    (executor.go:6449 mapReduce -> reduceFn).  An 8-byte all-reduce over xGMI is pure latency
    (tens of microseconds, comparable to a whole 1024-shard step), so the partial totals of
    consecutive steps are written into consecutive slots of a device vector and reduced together
    — the same bucketing data-parallel training applies to gradients — and two buckets alternate:
    the collective of bucket b runs asynchronously on the communicator's stream while the kernels
    of the following steps fill bucket b^1.  Every step's count is still reduced over all ranks;
    `flush()` completes the tail.  Works with any backend (RCCL for device tensors, gloo for CPU
    tensors in the tests).
    """

    def __init__(self, bucket: int, device=None):
        import torch

        self.bucket = int(bucket)
        self.buf = [torch.zeros(self.bucket, dtype=torch.int64, device=device) for _ in range(2)]
        self.work = [None, None]
        self.cur = 0  # bucket being filled
        self.fill = 0  # slots used in it
        self.collectives = 0

    def _free(self, b: int) -> None:
        if self.work[b] is not None:
            self.work[b].wait()  # stream-level wait for device tensors
            self.work[b] = None

    def slot(self):
        """The tensor view (1 element) the current step's partial total must be written to."""
        if self.fill == 0:
            self._free(self.cur)  # the previous collective on this bucket must have finished
            self.buf[self.cur].zero_()  # slots a partially filled tail bucket leaves unused must reduce to 0
        return self.buf[self.cur][self.fill : self.fill + 1]

    def slot_ptr(self) -> int:
        return self.slot().data_ptr()

    def _launch(self) -> None:
        import torch.distributed as dist

        b = self.cur
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.work[b] = dist.all_reduce(self.buf[b], op=dist.ReduceOp.SUM, async_op=True)
        self.collectives += 1
        self.cur ^= 1
        self.fill = 0

    def advance(self) -> None:
        """The current slot has been produced (enqueued); reduce the bucket when it is full."""
        self.fill += 1
        if self.fill == self.bucket:
            self._launch()

    def flush(self):
        """Reduce a partially filled bucket and wait for every outstanding collective.
        Returns the two buckets (reduced values of the last <= 2*bucket steps)."""
        if self.fill:
            self._launch()
        self._free(0)
        self._free(1)
        return self.buf
