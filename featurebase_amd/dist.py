"""Multi-GPU plumbing: one process per GPU, shards partitioned over ranks, partial counts
reduced with one collective (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests).

This mirrors the only cross-node exchange the reference's path has: `mapReduce` maps
per-shard functions on the node that owns each shard and folds count-valued results with
an associative `reduceFn` (executor.go:6449-6533, 5880); bitmap-valued results are never
exchanged (executor.go:1767).
"""
from __future__ import annotations

import os
import sys
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


def order_totals(totals: np.ndarray, n: int = 0):
    """(row indexes, counts) of a totals vector in the order TopN reports them: count descending, row index ascending inside
    one count (Pairs sorted by count, executor.go:2823; the tie rule is fbk_topn's), rows with a zero count dropped, the
    first n (0: all)."""
    t = np.asarray(totals, dtype=np.uint64)
    idx = np.nonzero(t)[0]
    idx = idx[np.lexsort((idx, -t[idx].astype(np.int64)))] if idx.size else idx
    if n:
        idx = idx[:n]
    return idx.astype(np.uint32), t[idx]


def topn_reduce(local_totals: np.ndarray, local_candidates: Optional[np.ndarray] = None, n: int = 0, device: Optional[str] = None):
    """TopN for the one-process-per-GPU deployment — the multi-process form of fbk_group_topn.  Every rank passes what
    fbk_topn_partials (Context.topn_partials) returned for the shards IT owns: local_totals[i] = its total of row i
    (thresholds applied per shard), local_candidates[i] != 0 iff one of its shards lists row i in the reference's first
    pass (fragment.top with N = n per SHARD, executor.go:2869-2944; a rank without shards passes zeros).  ONE all_reduce
    of [totals | candidates] (2 n_a words): in the reference a node returns its shards' merged pairs untrimmed
    (executeTopNShards :2829-2864), so the candidate set is the union over all shards wherever they live, and the second
    pass (:2812-2818) is the totals of those rows — which every rank already holds.  Rows no rank flagged are dropped, the
    rest ordered (count descending, row index ascending) and trimmed to n (:2823-2825).  local_candidates = None: no
    candidate pass (n = 0, or the exact semantics): one all_reduce of n_a words.  Identical on every rank, and identical
    to fbk_topn over all shards on one context."""
    import torch.distributed as dist

    t = np.ascontiguousarray(local_totals, dtype=np.uint64)
    c = None if local_candidates is None else np.ascontiguousarray(local_candidates, dtype=np.uint64)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if multi:
        # ALWAYS the same collective on every rank — [totals | candidates | 1 if this rank passed candidates] — so that ranks that
        # disagree about the candidate pass fail together instead of hanging in all_reduces of different lengths
        flag = np.array([0 if c is None else 1], dtype=np.uint64)
        both = reduce_count_vector(np.concatenate([t, np.zeros_like(t) if c is None else c, flag]), device)
        with_c = int(both[-1])
        if with_c not in (0, dist.get_world_size()):
            raise ValueError(f"topn_reduce: {with_c} of {dist.get_world_size()} ranks passed candidate flags (every rank must use the same topn_semantics)")
        t, c = both[: t.size], (both[t.size:-1] if with_c else None)
    if c is not None:
        t = np.where(c != 0, t, np.uint64(0))
    return order_totals(t, n)


def bsi_sum_reduce(psum: int, nsum: int, count: int, device: Optional[str] = None):
    """Sum(field) over all ranks (executeSum's reduce, ValCount.Add, executor.go:8438): every rank passes the {psum, nsum,
    count} of the shards it owns (fbk_bsi_sum folded over its shards; uint64 sums wrap exactly as the reference's); returns
    (int64(psum) - int64(nsum) with that wrap-around, count) — roaring/filter.go:1103-1108.  The caller adds count * Base."""
    v = reduce_count_vector(np.array([psum, nsum, count], dtype=np.uint64), device)
    s = (int(v[0]) - int(v[1])) & 0xFFFFFFFFFFFFFFFF
    return (s - (1 << 64) if s >= (1 << 63) else s), int(v[2])


def _filled(t):
    """`t` (a tensor just created by torch.zeros) with its fill COMPLETE.  The fill is a kernel on the stream that was current when
    the tensor was made — usually torch's default stream —, while the cells are written on the fbk context's stream and read by
    the collective: without this wait a late fill can wipe a cell between the count kernel and the all-reduce (seen once with
    eight ranks sharing one GPU: every rank's reduced totals short by one rank's share)."""
    if t.is_cuda:
        import torch

        torch.cuda.current_stream(t.device).synchronize()
    return t


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
        self.buf = [_filled(torch.zeros(self.bucket, dtype=torch.int64, device=device)) for _ in range(2)]
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


class PerQueryReducer:
    """One collective per query, pipelined on the device: what a single query pays for its exchange step.

    A query's partial result (1 count, or the n_a x n_b cells of a GroupBy matrix: mergeGroupCounts,
    executor.go:3728-3762) is written by the rank's kernels into a CELL of `width` int64 on the device and
    all-reduced on its own — no bucketing over queries.  `depth` cells rotate so that the kernels of query
    k + 1 never write the buffer the collective of query k is still reading: `cell()` hands out the next one
    after (stream-)waiting for the collective that last used it, `reduce()` starts the asynchronous
    all-reduce of the cell just written.  With no process group (N = 1) nothing is exchanged.

    STREAM RULE.  The collective is ordered against torch's CURRENT stream only.  The kernels that write the cell
    run on the fbk context's stream, so either (a) the context runs on torch's current stream
    (`ctx.set_stream(torch.cuda.current_stream().cuda_stream)` inside `with torch.cuda.stream(...)`, what bench.py
    does), or (b) pass the producer's stream as `producer_stream` (a torch.cuda.Stream / ExternalStream wrapping
    the context's stream): `reduce()` then records an event on it and makes the current stream wait for that event
    before the all-reduce is enqueued.  With neither, the collective may read a cell before it is written."""

    def __init__(self, width: int, depth: int, device=None, producer_stream=None, always: bool = False):
        import torch

        if int(width) < 1 or int(depth) < 1:
            raise ValueError("PerQueryReducer: width and depth must be >= 1")
        self.producer_stream = producer_stream
        self.always = bool(always)  # issue the collective even in a one-rank group (scripts/collective_host_cost.py: what a call costs the launching thread)
        self.width, self.depth = int(width), int(depth)
        self.buf = _filled(torch.zeros((self.depth, self.width), dtype=torch.int64, device=device))
        self.work = [None] * self.depth
        self.k = 0
        self.collectives = 0

    def cell(self):
        i = self.k % self.depth
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return self.buf[i]

    def reduce(self):
        import torch.distributed as dist

        i = self.k % self.depth
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.always):
            if self.producer_stream is not None and self.buf.is_cuda:
                import torch

                ev = torch.cuda.Event()
                ev.record(self.producer_stream)
                torch.cuda.current_stream(self.buf.device).wait_event(ev)
            self.work[i] = dist.all_reduce(self.buf[i], op=dist.ReduceOp.SUM, async_op=True)
            self.collectives += 1
        self.k += 1
        return i

    def flush(self):
        for i in range(self.depth):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
        return self.buf


def library_comm_init(ctx) -> bool:
    """Give the fbk context `ctx` of THIS rank a communicator of its own over all ranks of the torch process group
    (fbk_comm_*): rank 0 draws the unique id, torch broadcasts the 128 bytes — its only part — and every rank enters
    ncclCommInitRank.  True if EVERY rank succeeded (agreed through one all-reduce); False leaves no communicator behind,
    and the caller stays on torch's collectives."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    box = [None]
    ok = 1
    try:
        if rank == 0:
            box[0] = ctx.comm_unique_id()
    except Exception as e:  # noqa: BLE001 — no librccl in this process: every rank learns it below
        print(f"[fbk dist] library communicator unavailable: {e}", file=sys.stderr, flush=True)
        box[0] = b""
    dist.broadcast_object_list(box, src=0)
    if not box[0]:
        return False
    try:
        ctx.comm_init(box[0], world, rank)
    except Exception as e:  # noqa: BLE001
        print(f"[fbk dist] rank {rank}: fbk_comm_init failed: {e}", file=sys.stderr, flush=True)
        ok = 0
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else None
    t = torch.tensor([ok], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) == 0:
        if ok:
            ctx.comm_close()
        return False
    # Pre-flight: ONE all-reduce through the new communicator, checked on every rank against the closed form.  The library's
    # collectives have never run over more than one device; a wrong sum here sends every rank back to torch's collectives
    # instead of failing the run's parity assert later.
    good = 1
    try:
        probe = torch.tensor([rank + 1], dtype=torch.int64, device=dev)
        if dev is not None:
            torch.cuda.synchronize()
        ctx.comm_all_reduce(probe.data_ptr(), 1)
        ctx.comm_fence()
        if dev is not None:
            torch.cuda.synchronize()
        if int(probe.item()) != world * (world + 1) // 2:
            print(f"[fbk dist] rank {rank}: the library's all-reduce returned {int(probe.item())}, expected {world * (world + 1) // 2}", file=sys.stderr, flush=True)
            good = 0
    except Exception as e:  # noqa: BLE001
        print(f"[fbk dist] rank {rank}: the library's all-reduce failed: {e}", file=sys.stderr, flush=True)
        good = 0
    t = torch.tensor([good], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) == 0:
        ctx.comm_close()
        return False
    return True


class LibraryPerQueryReducer:
    """PerQueryReducer with the collective issued by the LIBRARY (fbk_comm_all_reduce_u64: RCCL through the library's own
    communicator, on the communicator's stream, ordered after the context's stream by one event) instead of through
    torch.distributed — whose call path costs the launching thread ~28 us per all-reduce, more than half of a 41 us headline
    step (profiles/r06_collective_host_cost.json).  Same cells, same rotation: `cell()` hands out the next cell, `reduce()`
    starts the all-reduce of the cell just written, `fence()` (once per revolution, before the ring is cleared, and before
    the cells are read) makes the context's stream wait for every collective so far — no host wait anywhere."""

    def __init__(self, ctx, width: int, depth: int, device=None):
        import torch

        if int(width) < 1 or int(depth) < 1:
            raise ValueError("LibraryPerQueryReducer: width and depth must be >= 1")
        self.ctx, self.width, self.depth = ctx, int(width), int(depth)
        self.buf = _filled(torch.zeros((self.depth, self.width), dtype=torch.int64, device=device))
        self.base = self.buf.data_ptr()
        self.k = 0
        self.collectives = 0

    def cell_ptr(self) -> int:
        return self.base + (self.k % self.depth) * self.width * 8

    def reduce(self) -> int:
        i = self.k % self.depth
        self.ctx.comm_all_reduce(self.base + i * self.width * 8, self.width)
        self.collectives += 1
        self.k += 1
        return i

    def flush(self):
        self.ctx.comm_fence()
        return self.buf


def host_add(partial, pinned, cpu_group=None):
    """"Copy the partials to the host and add": the alternative to the device collective.
    `partial` is this rank's device (or CPU) tensor, `pinned` a host tensor of the same shape; returns `pinned`
    holding the sum over ranks (gloo all-reduce on the host)."""
    import torch
    import torch.distributed as dist

    pinned.copy_(partial, non_blocking=True)
    if partial.is_cuda:
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(pinned, op=dist.ReduceOp.SUM, group=cpu_group)
    return pinned


def strong_scaling_queries(run_local, width: int, n_queries: int, device, expected=None, sync=None, depth: int = 8, cpu_group=None, warmup: int = 2):
    """The multi-GPU query loop of bench.py's strong-scaling section, independent of what computes the partials:
    `run_local(cell)` enqueues this rank's partial result (width int64) into `cell`.  Measures, over n_queries
    queries each: (a) one device all-reduce per query, pipelined (throughput of independent queries, no query
    amortises another's collective); (b) the same with the result read back by the host after every query (the
    latency one query sees); (c) host add.  `expected` (numpy uint64 [width], the sum over all ranks) is checked
    in every mode.  Returns a dict of seconds per query (this rank's clock; the caller takes the max over ranks)."""
    import time

    import torch
    import torch.distributed as dist

    if n_queries < 1:
        raise ValueError("strong_scaling_queries: n_queries must be >= 1")
    sync = sync or (lambda: None)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def barrier():
        if multi:
            dist.barrier()

    def check(t, what):
        if expected is not None:
            got = t.detach().cpu().numpy().view(np.uint64).reshape(-1)
            assert (got == np.asarray(expected, dtype=np.uint64).reshape(-1)).all(), f"{what}: reduced result differs from the expected total"

    red = PerQueryReducer(width, depth, device)
    for _ in range(warmup):
        run_local(red.cell())
        red.reduce()
    red.flush()
    sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_queries):
        run_local(red.cell())
        red.reduce()
    bufs = red.flush()
    sync()
    barrier()
    pipelined = (time.perf_counter() - t0) / n_queries
    for i in range(min(depth, n_queries)):
        check(bufs[i], "collective per query")
    lat = []
    for _ in range(n_queries):
        t1 = time.perf_counter()
        c = red.cell()
        run_local(c)
        i = red.reduce()
        red.flush()
        sync()
        host = bufs[i].cpu()
        lat.append(time.perf_counter() - t1)
    check(host, "collective per query, read back")
    pinned = torch.zeros(width, dtype=torch.int64)
    if device is not None and torch.device(device).type == "cuda":
        pinned = pinned.pin_memory()
    cell = _filled(torch.zeros(width, dtype=torch.int64, device=device))
    hl = []
    for _ in range(n_queries):
        t1 = time.perf_counter()
        run_local(cell)
        sync()
        host_add(cell, pinned, cpu_group)
        hl.append(time.perf_counter() - t1)
    check(pinned, "host add")
    lat.sort()
    hl.sort()
    return {"pipelined_s_per_query": pipelined, "latency_s": {"median": lat[len(lat) // 2], "p10": lat[len(lat) // 10], "p90": lat[(len(lat) * 9) // 10]},
            "host_add_latency_s": {"median": hl[len(hl) // 2], "p10": hl[len(hl) // 10], "p90": hl[(len(hl) * 9) // 10]}, "collectives": red.collectives, "queries": n_queries}
