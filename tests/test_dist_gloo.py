"""world_size-2 test of the multi-GPU path on CPU (gloo): shard partition + one all-reduce of
the partial count vectors reproduces the single-process result.  The per-shard partial
counts come from the oracle here (no GPU in this container); on the GPU box the same
`featurebase_amd.dist` functions carry the counts the HIP kernels produce (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank: int, world: int, port: int, n_shards: int, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import datagen as D
    from featurebase_amd import dist as fd
    from oracle import pyoracle as O

    fd.init("gloo")
    mine = fd.shards_for_rank(n_shards, rank, world)
    # Count(Intersect(a, b)): per-shard |a ∩ b| (scalar reduce) and a 2x2 GroupBy-style matrix
    total = np.zeros(1, dtype=np.uint64)
    mat = np.zeros(4, dtype=np.uint64)
    for s in mine:
        rng = D.rng_for(900 + s)
        rows = [O.OBitmap.from_containers(list(D.random_row(rng, 0).items())) for _ in range(4)]
        total[0] += rows[0].intersection_count(rows[1])
        for i in range(2):
            for j in range(2):
                mat[i * 2 + j] += rows[i].intersection_count(rows[2 + j])
    total = fd.reduce_count_vector(total)
    mat = fd.reduce_count_vector(mat)
    q.put((rank, mine, int(total[0]), mat.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partition_is_a_partition():
    sys.path.insert(0, ROOT)
    from featurebase_amd import dist as fd

    for world in (1, 2, 4, 8):
        parts = [fd.shards_for_rank(8192, r, world) for r in range(world)]
        flat = sorted(s for p in parts for s in p)
        assert flat == list(range(8192))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.timeout(300)
def test_two_ranks_reduce_to_single_process_result():
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datagen as D
    from oracle import pyoracle as O

    O.build()
    n_shards, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_shards, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process expectation
    exp_total, exp_mat = 0, [0, 0, 0, 0]
    for s in range(n_shards):
        rng = D.rng_for(900 + s)
        rows = [O.OBitmap.from_containers(list(D.random_row(rng, 0).items())) for _ in range(4)]
        exp_total += rows[0].intersection_count(rows[1])
        for i in range(2):
            for j in range(2):
                exp_mat[i * 2 + j] += rows[i].intersection_count(rows[2 + j])
    seen = []
    for rank, mine, total, mat in results:
        assert total == exp_total and mat == exp_mat, (rank, total, exp_total)
        seen += mine
    assert sorted(seen) == list(range(n_shards))


def _bucket_worker(rank: int, world: int, port: int, steps: int, bucket: int, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from featurebase_amd import dist as fd

    fd.init("gloo")
    red = fd.BucketedCountReducer(bucket)
    seen = []
    for k in range(steps):
        if red.fill == 0 and k >= 2 * bucket:
            # the bucket about to be refilled holds reduced values of steps k-2*bucket .. k-bucket-1
            red._free(red.cur)
            seen += red.buf[red.cur].tolist()
        slot = red.slot()  # bench.py hands slot_ptr() to fbk_plan_total; here the "kernel" is a host write
        slot[0] = (rank + 1) * 1000 + k  # this rank's partial total of step k
        red.advance()
    bufs = red.flush()
    q.put((rank, seen, [b.tolist() for b in bufs], red.collectives))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_async_reduce_two_ranks():
    """bench.py's N > 1 reduce: per-step partial totals collected in two alternating buckets, one
    asynchronous all-reduce per bucket, tail flushed; every step's total is the sum over ranks."""
    import torch.multiprocessing as mp

    steps, bucket, world = 37, 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, steps, bucket, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    expect = lambda k: sum((r + 1) * 1000 + k for r in range(world))  # noqa: E731
    for rank, seen, bufs, collectives in results:
        assert collectives == (steps + bucket - 1) // bucket
        # buckets recycled during the run held the reduced totals of complete earlier buckets
        assert seen == [expect(k) for k in range(len(seen))] and len(seen) == (steps // bucket - 2 + (1 if steps % bucket else 0)) * bucket
        # after flush: the last full bucket and the partially filled tail bucket
        tail = steps % bucket
        last_full_start = (steps // bucket - 1) * bucket
        flat = sorted(v for b in bufs for v in b if v)
        assert flat == sorted([expect(k) for k in range(last_full_start, last_full_start + bucket)] + [expect(k) for k in range(steps - tail, steps)])


def test_bucketed_reducer_without_a_process_group():
    """N = 1 (bench.py at --gpus 1): no collective, the buckets are only cleared on reuse and
    every step's slot keeps what was accumulated into it."""
    import torch

    from featurebase_amd.dist import BucketedCountReducer

    red = BucketedCountReducer(4, torch.device("cpu"))
    seen = []
    for step in range(10):
        slot = red.slot()
        assert int(slot.item()) == 0  # cleared before reuse
        slot += 100 + step  # what the accumulate kernel does on the device
        seen.append(100 + step)
        red.advance()
    bufs = red.flush()
    assert red.collectives == 3  # 4 + 4 + tail of 2
    vals = torch.cat(bufs).tolist()
    # the last two buckets hold steps 4..7 and 8..9 (+ two cleared slots)
    assert sorted(v for v in vals if v) == seen[4:]


def _strong_worker(rank: int, world: int, port: int, n_shards: int, q):
    """bench.py's strong-scaling section (config 4: GroupBy matrix over a FIXED number of shards split over the
    ranks) with the oracle standing in for the GPU: the same featurebase_amd.dist.strong_scaling_queries loop —
    per-query collectives through the PerQueryReducer ring, the host-readback latency mode and host add."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import datagen as D
    from featurebase_amd import dist as fd
    from oracle import pybatch as PB

    fd.init("gloo")
    n_a = n_b = 4
    mine = fd.shards_for_rank(n_shards, rank, world)

    def shard_rows(s):
        return D.dense_rows(n_a + n_b + 1, 0.5, 4400 + s)  # rows 0..3 field A, 4..7 field B, 8 the filter

    def matrix_of(shards):
        w = np.concatenate([shard_rows(s) for s in shards]) if shards else np.zeros((0, 16, 1024), dtype=np.uint64)
        if not shards:
            return np.zeros((n_a, n_b), dtype=np.uint64)
        rs = PB.RowSet.from_dense(w)
        base = np.arange(len(shards))[:, None] * (n_a + n_b + 1)
        return PB.count_matrix(rs, base + np.arange(n_a), rs, base + n_a + np.arange(n_b), rs, (base + n_a + n_b).reshape(-1), nthreads=1).sum(axis=0)

    partial = torch.from_numpy(matrix_of(mine).view(np.int64).reshape(-1).copy())
    expected = matrix_of(list(range(n_shards))).reshape(-1)
    calls = [0]

    def run_local(cell):
        cell.copy_(partial)  # the kernels' job on the GPU box: this rank's partial matrix into the cell
        calls[0] += 1

    res = fd.strong_scaling_queries(run_local, n_a * n_b, 11, None, expected=expected, depth=4)
    q.put((rank, mine, res, calls[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_strong_scaling_query_loop_two_ranks():
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    from oracle import pyoracle as O

    O.build()
    n_shards, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, world, port, n_shards, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)  # (every mode checked its reduced matrix against the all-shard oracle result)
    seen = []
    for rank, mine, res, calls in results:
        seen += mine
        assert res["queries"] == 11 and res["collectives"] == 2 + 11 + 11  # warm-up + pipelined + read-back: one collective per query, never bucketed
        assert calls == 2 + 3 * 11
        assert set(res) >= {"pipelined_s_per_query", "latency_s", "host_add_latency_s"} and res["latency_s"]["median"] > 0
    assert sorted(seen) == list(range(n_shards))


def test_per_query_reducer_without_a_process_group():
    import torch

    from featurebase_amd.dist import PerQueryReducer

    red = PerQueryReducer(3, 2, torch.device("cpu"))
    for k in range(5):
        c = red.cell()
        c.copy_(torch.tensor([k, k + 1, k + 2]))
        assert red.reduce() == k % 2
    bufs = red.flush()
    assert red.collectives == 0 and bufs.tolist() == [[4, 5, 6], [3, 4, 5]]


def _topn_worker(rank: int, world: int, port: int, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from featurebase_amd import dist as fd
    from oracle import pytopn as T

    fd.init("gloo")
    out = []
    for case in range(12):
        rng = np.random.default_rng(3100 + case)
        n_rows, n_shards = int(rng.integers(3, 40)), int(rng.integers(1, 9))
        # shard s: row id -> columns; src = the filter row of the shard.  Every rank builds ALL shards (same seed) and owns s mod world
        shards = [{r: sorted(set(rng.integers(0, 64, int(rng.integers(0, 30))).tolist())) for r in range(n_rows)} for _ in range(n_shards)]
        srcs = [sorted(set(rng.integers(0, 64, int(rng.integers(1, 40))).tolist())) for _ in range(n_shards)]
        mt = int(rng.integers(0, 4)) if case % 3 == 0 else 0
        tt = int(rng.choice([10, 30])) if case % 4 == 1 else 0
        n = int(rng.choice([0, 1, 2, 5, n_rows, n_rows + 3]))
        ids = list(range(n_rows))
        mine = fd.shards_for_rank(n_shards, rank, world)
        # what fbk_topn_partials returns for this rank's shards (tests/test_gpu_topn.py checks the device against exactly this)
        local, cand = np.zeros(n_rows, dtype=np.uint64), np.zeros(n_rows, dtype=np.uint64)
        if mine:
            for r, c in T.top_exact([shards[s] for s in mine], ids, 0, [srcs[s] for s in mine], mt, tt):
                local[r] = c
            cand[T.topn_candidates([shards[s] for s in mine], n, [srcs[s] for s in mine], mt, tt)] = 1
        idx, cnt = fd.topn_reduce(local, cand, n)
        exp = T.execute_topn(shards, n, srcs, None, mt, tt)  # ALL shards, wherever they live
        exact_idx, exact_cnt = fd.topn_reduce(local, None, n)
        exact = T.top_exact(shards, ids, n, srcs, mt, tt)
        out.append(([(int(i), int(c)) for i, c in zip(idx, cnt)], [(int(r), int(c)) for r, c in exp]))
        out.append(([(int(i), int(c)) for i, c in zip(exact_idx, exact_cnt)], [(int(r), int(c)) for r, c in exact]))
    # BSI Sum: {psum, nsum, count} per rank, with a negative total and a wrap-around of the uint64 partial sums
    parts = [(5, 1 << 63, 3), ((1 << 64) - 7, (1 << 63) + 10, 4)]
    s, c = fd.bsi_sum_reduce(*parts[rank])
    # ranks that disagree about the candidate pass: the same collective on both, then the same error on both (no hang)
    try:
        fd.topn_reduce(np.ones(4, dtype=np.uint64), np.ones(4, dtype=np.uint64) if rank == 0 else None, 2)
        out.append(("no error", "ValueError"))
    except ValueError:
        pass
    q.put((rank, out, (s, c)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_topn_reduce_and_bsi_sum_two_ranks():
    """featurebase_amd.dist.topn_reduce / bsi_sum_reduce (the one-process-per-GPU forms of fbk_group_topn / fbk_group_bsi_sum)
    on two gloo ranks against oracle/pytopn.execute_topn over ALL shards (executeTopN, executor.go:2779-2864: per-SHARD
    candidates, so the answer does not depend on which rank owns which shard) and ValCount.Add's arithmetic; without the
    candidate flags the reduce gives the exact top n (top_exact)."""
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_topn_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    total = (5 + (1 << 64) - 7) % (1 << 64) - ((1 << 63) + (1 << 63) + 10) % (1 << 64)
    total &= (1 << 64) - 1
    total = total - (1 << 64) if total >= (1 << 63) else total
    for rank, cases, (s, c) in results:
        for got, exp in cases:
            assert got == exp, (rank, got, exp)
        assert (s, c) == (total, 7)


def test_topn_reduce_without_a_process_group():
    sys.path.insert(0, ROOT)
    from featurebase_amd import dist as fd

    idx, cnt = fd.topn_reduce(np.array([3, 0, 9, 3, 1], dtype=np.uint64), None, 3)
    assert idx.tolist() == [2, 0, 3] and cnt.tolist() == [9, 3, 3]  # count descending, row index ascending inside a count, zeros dropped
    idx, cnt = fd.topn_reduce(np.array([3, 0, 9, 3, 1], dtype=np.uint64), np.array([1, 1, 0, 0, 1], dtype=np.uint64), 3)
    assert idx.tolist() == [0, 4] and cnt.tolist() == [3, 1]  # rows 2 and 3 are nobody's candidates
    idx, cnt = fd.topn_reduce(np.zeros(4, dtype=np.uint64), None, 0)
    assert idx.size == 0 and cnt.size == 0
    assert fd.bsi_sum_reduce(7, 9, 2) == (-2, 2)
