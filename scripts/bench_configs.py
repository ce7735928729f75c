"""Secondary measurements: BASELINE.json configs 3, 4 (per-GPU slice) and 5 on one MI355X.
Not the headline (bench.py is); prints one JSON line per config with the algorithmic bytes,
GPU time (HIP events on the library's stream) and achieved GB/s.  Run it under
`rocprofv3 --kernel-trace --stats` to get per-kernel durations.

    python scripts/bench_configs.py [--shards3 64] [--shards4 128] [--iters 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import datagen as D  # noqa: E402
from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402


CPU_BASELINE = False
ORACLE_ROWS3 = {}


def timed(stream, fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(stream)
        fn()
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e-3  # median, seconds


def encoded_bytes(c):
    return {1: 2 * c.length, 2: 8192, 3: 4 * c.length}[c.typ]


def cpu_time(fn, min_s=1.0):
    """seconds per call of fn on one host thread (repeated until min_s has elapsed)"""
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_s:
            return dt / n


def config3(ctx, stream, n_shards, iters):
    """mixed array/run/bitmap containers, rank-law density 0.001..0.5, Union-of-64 rows then
    IntersectionCount against a filter row (fused: the union never touches HBM)."""
    k = 64
    rows, groups, nbytes, ncont = [], [], 0, 0
    t0 = time.time()
    for s in range(n_shards):
        rng = D.rng_for(3000 + s)
        ids = []
        for r in range(k):
            d = D.zipf_density(r)
            row = {}
            for slot in range(16):
                rs = rng.random() < 0.25
                c = D.fbk_container_of_vals(D.mixed_vals_for_density(rng, d, rs))  # numpy only: no oracle code while measuring
                if c is not None and c.n:
                    row[s * 16 + slot] = c
                    nbytes += encoded_bytes(c)
                    ncont += 1
                    if s == 0:
                        ORACLE_ROWS3.setdefault(r, []).append((slot, c))
            ids.append(len(rows))
            rows.append(row)
        groups.append(ids)
    frows = []
    for s in range(n_shards):
        rng = D.rng_for(3500 + s)
        row = {}
        for slot in range(16):
            c = D.fbk_container_of_vals(D.mixed_vals_for_density(rng, 0.5, False))
            row[s * 16 + slot] = c
            nbytes += encoded_bytes(c)
            ncont += 1
            if s == 0:
                ORACLE_ROWS3.setdefault("filter", []).append((slot, c))
        frows.append(row)
    gen_s = time.time() - t0
    batch, F = ctx.upload(rows), ctx.upload(frows)
    groups = np.array(groups, dtype=np.uint32)
    fidx = np.arange(n_shards)
    t = timed(stream, lambda: ctx.union_n_intersection_count(batch, groups, F, fidx), iters)
    tm = timed(stream, lambda: ctx.union_n(batch, groups)[0].free(), max(2, iters // 2))
    # TopN / TopK shape on the same rows: |row_r ∩ filter| for all 64 rows of every shard
    t_topn = timed(stream, lambda: ctx.count_matrix(batch, groups, F, fidx.reshape(-1, 1)), iters)
    t_topk = timed(stream, lambda: ctx.topk(batch, groups, 10, F, fidx), iters)  # + reduce over shards + ordering on the device
    pair_a = groups.reshape(-1)
    pair_f = np.repeat(fidx, k)
    t_pairs = timed(stream, lambda: ctx.intersection_count(batch, pair_a, F, pair_f), iters)
    # GroupBy shape on the same mixed rows: rows 0..31 x rows 32..63 of every shard, with the filter
    t_gb = timed(stream, lambda: ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, fidx), max(3, iters // 2))
    set_ops = n_shards * 16 * k  # (k-1) unions + 1 intersection count per slot
    cpu = None
    if CPU_BASELINE:
        from oracle import pyoracle as O

        def to_oracle(c):  # the same encoding and bytes, as an oracle container (only in this --cpu-baseline leg)
            return O.OContainer.array(c.data) if c.typ == O.ARRAY else O.OContainer.run(c.data.tolist()) if c.typ == O.RUN else O.OContainer.bitmap(c.data, c.n)

        bms = [O.OBitmap.from_containers([(sl, to_oracle(c)) for sl, c in ORACLE_ROWS3.get(r, [])]) for r in range(k)]
        fb = O.OBitmap.from_containers([(sl, to_oracle(c)) for sl, c in ORACLE_ROWS3["filter"]])
        t_cpu = cpu_time(lambda: bms[0].union(*bms[1:]).intersection_count(fb))  # Bitmap.Union n-way + IntersectionCount
        cpu = {"per_shard_s_1thread": t_cpu, "shards_per_s_1thread": 1 / t_cpu, "kind": "port",
               "what": "oracle Bitmap.Union(63 others) + IntersectionCount(filter) of shard 0, one host thread"}
    return {
        "cpu_baseline": cpu,
        "config": 3, "workload": f"{n_shards} shards x (64 rows + filter), mixed containers, Union-of-64 then IntersectionCount (fused)",
        "containers": ncont, "algorithmic_bytes": nbytes, "gpu_s": t, "GBps": nbytes / t / 1e9, "set_ops_per_s": set_ops / t,
        "materialised_union_gpu_s": tm, "host_gen_s": gen_s,
        "groupby_32x32_mixed_gpu_s": t_gb, "topn_count_matrix_gpu_s": t_topn, "topk10_gpu_s": t_topk, "topn_pairs_gpu_s": t_pairs,
        "topn_GBps": nbytes / min(t_topn, t_pairs) / 1e9, "topn_note": "64 rows x 1 filter row per shard (doTopK shape); bytes = every container once",
    }


def config4(ctx, stream, n_shards, iters, n_a=32, n_b=32):
    """GroupBy/TopN-style many-row IntersectionCount matrix with a filter row, dense bitmaps
    (upper bound on bytes per shard: 65 rows x 128 KiB)."""
    wa = D.dense_rows(n_shards * n_a, 0.5, 4001)
    wb = D.dense_rows(n_shards * n_b, 0.5, 4002)
    wf = D.dense_rows(n_shards, 0.5, 4003)
    A, B, F = ctx.upload_dense(wa), ctx.upload_dense(wb), ctx.upload_dense(wf)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)
    t = timed(stream, lambda: ctx.count_matrix(A, ra, B, rb, F, rf), iters)
    # spot check one cell against numpy
    tot = ctx.count_matrix(A, ra, B, rb, F, rf)
    exp = int(sum(np.bitwise_count(wa[s * n_a + 3] & wb[s * n_b + 5] & wf[s]).sum() for s in range(n_shards)))
    assert int(tot[3, 5]) == exp
    world = int(os.environ.get("WORLD_SIZE", "1"))
    t_reduce = None
    if world > 1:  # mergeGroupCounts across nodes (executor.go:3728): one all-reduce of the nA x nB matrix
        from featurebase_amd import dist as fdist

        dev = torch.device("cuda", torch.cuda.current_device())
        t0 = time.perf_counter()
        glob = fdist.reduce_count_vector(tot.reshape(-1), dev)
        torch.cuda.synchronize()
        t_reduce = time.perf_counter() - t0
        assert int(glob.reshape(n_a, n_b)[3, 5]) == exp * world  # every rank generated the same shards here
    nbytes = n_shards * (n_a + n_b + 1) * 16 * 8192
    # TopN shape on dense rows: the n_a rows of A against the single filter row per shard
    t_topn = timed(stream, lambda: ctx.count_matrix(A, ra, F, rf.reshape(-1, 1)), iters)
    topn_bytes = n_shards * (n_a + 1) * 16 * 8192
    # n-way union of the same dense rows, fused with |union ∩ filter|
    t_union = timed(stream, lambda: ctx.union_n_intersection_count(A, ra, F, rf), iters)
    cpu = None
    if CPU_BASELINE:
        from oracle import pybsi as PB
        from oracle import pyoracle as O

        mk = lambda w: O.OBitmap.from_containers([(sl, O.OContainer.bitmap(np.asarray(w).reshape(16, 1024)[sl])) for sl in range(16)])  # noqa: E731
        fa = PB.Fragment([mk(wa[i]) for i in range(n_a)])
        fb = PB.Fragment([mk(wb[j]) for j in range(n_b)])
        ff = mk(wf[0])
        t_cpu = cpu_time(lambda: PB.groupby_counts(fa, fb, ff))
        cpu = {"per_shard_s_1thread": t_cpu, "shards_per_s_1thread": 1 / t_cpu, "kind": "port",
               "what": "oracle groupByIterator counts (32 x 32 rows + filter) of shard 0, one host thread"}
    return {
        "cpu_baseline": cpu,
        "topn_dense_gpu_s": t_topn, "topn_dense_GBps": topn_bytes / t_topn / 1e9,
        "union_dense_gpu_s": t_union, "union_dense_GBps": topn_bytes / t_union / 1e9,
        "world_size": world, "matrix_allreduce_s": t_reduce,
        "config": 4, "workload": f"{n_shards} shards x ({n_a} x {n_b} rows + filter), dense bitmaps, count matrix",
        "algorithmic_bytes_read_once": nbytes, "gpu_s": t, "GBps_vs_read_once": nbytes / t / 1e9,
        "set_ops_per_s": n_shards * 16 * n_a * n_b / t, "pair_bits_scanned_GBps": n_shards * n_a * n_b * 2 * 16 * 8192 / t / 1e9,
    }


def config5(ctx, stream, iters, n_shards=96, depth=64):
    """BSI Range(> k) + Sum over 64 bit planes + exists + sign (dense planes)."""
    w = D.dense_rows(n_shards * (depth + 2), 0.5, 5001)
    w = w.reshape(n_shards, depth + 2, 16, 1024)
    w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)  # exists: every column has a value
    w[-1, 0, 6:] = 0  # last shard partial (100M columns = 95 full shards + 385 280 columns)
    batch = ctx.upload_dense(w.reshape(-1))
    base = np.arange(n_shards, dtype=np.uint32) * (depth + 2)
    k = 1 << 62
    t_range = timed(stream, lambda: ctx.bsi_range(batch, base, L.BSI_GT, depth, k)[0].free(), iters)
    out, cnt = ctx.bsi_range(batch, base, L.BSI_GT, depth, k)
    t_sum_f = timed(stream, lambda: ctx.bsi_sum(batch, base, depth, out, np.arange(n_shards)), iters)
    t_sum = timed(stream, lambda: ctx.bsi_sum(batch, base, depth), iters)
    out.free()
    plane_bytes = n_shards * 16 * 8192
    cpu = None
    if CPU_BASELINE:
        from oracle import pybsi as PB
        from oracle import pyoracle as O

        fr = PB.Fragment([O.OBitmap.from_containers([(sl, O.OContainer.bitmap(w[0, r, sl])) for sl in range(16)]) for r in range(depth + 2)])
        t_sum_cpu = cpu_time(lambda: PB.bsi_sum(fr, None, False))
        t_rng_cpu = cpu_time(lambda: PB.bsi_range(fr, PB.GT, depth, k))
        cpu = {"sum_per_shard_s_1thread": t_sum_cpu, "range_per_shard_s_1thread": t_rng_cpu, "kind": "port",
               "what": "oracle fragment.sum / fragment.rangeOp(GT) of shard 0 (66 dense rows), one host thread"}
    return {
        "cpu_baseline": cpu,
        "config": 5, "workload": f"BSI {n_shards} shards x (64 planes + exists + sign), dense; Range(>2^62), Sum(filter=range), Sum",
        "range_gpu_s": t_range, "range_GBps": plane_bytes * (depth + 2 + 1) / t_range / 1e9,
        "range_note": "Range(> 2^62) reads exists, sign and all 64 planes once, writes 1 row",
        "sum_filtered_gpu_s": t_sum_f, "sum_filtered_GBps": plane_bytes * (depth + 3) / t_sum_f / 1e9,
        "sum_gpu_s": t_sum, "sum_GBps": plane_bytes * (depth + 2) / t_sum / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards3", type=int, default=64)
    ap.add_argument("--shards4", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=int, default=0)
    ap.add_argument("--cpu-baseline", action="store_true", help="also time the CPU oracle on one shard of each config (one host thread)")
    a = ap.parse_args()
    global CPU_BASELINE
    CPU_BASELINE = a.cpu_baseline
    # one process per GPU (torchrun): every rank holds its own shards; count-valued results are
    # summed over ranks with one RCCL all-reduce (config 4's "partial-count reduce over xGMI")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FBK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        from featurebase_amd import dist as fdist

        fdist.init(backend, torch.device("cuda", dev_index))
    ctx = Context(dev_index)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        rank0 = int(os.environ.get("RANK", "0")) == 0
        if a.only in (0, 4):
            r = config4(ctx, stream, a.shards4, a.iters)
            if rank0:
                print(json.dumps(r), flush=True)
        if a.only in (0, 5) and world == 1:
            print(json.dumps(config5(ctx, stream, a.iters)), flush=True)
        if a.only in (0, 3) and world == 1:
            print(json.dumps(config3(ctx, stream, a.shards3, a.iters)), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
