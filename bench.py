#!/usr/bin/env python3
"""bench.py — headline benchmark of the roaring set-op hot path on MI355X.

Workload (BASELINE.json configs[1]): per GPU 1024 shards x 2 rows x 2^20 columns, density
50 % => 32768 bitmap containers (256 MiB) resident in HBM; one *step* = one pass of
Count(Intersect(Row a, Row b)) over every shard: |a_s ∩ b_s| for each shard s
(Bitmap.IntersectionCount, roaring.go:711), the per-node sum (executeCount reduceFn,
executor.go:5880) and, for N > 1, the cross-GPU sum of the partial counts by one RCCL
all-reduce (the exchange step executor.go:6449 mapReduce does over HTTP).

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself
(torch.distributed.run, one rank per GPU, RCCL; when the box has fewer than N GPUs the ranks share
the devices and reduce over gloo — flagged as "oversubscribed" in the output).  Under torchrun it
reads RANK / LOCAL_RANK / WORLD_SIZE as usual.  Prints ONE JSON line on rank 0.

`value` = container set-ops per second over all N GPUs (a set-op = one container pair with equal
keys); weak scaling (1024 shards per GPU).  For N > 1 EVERY step carries its own RCCL
all-reduce of its partial total (one collective per query, pipelined on the device — nothing is amortised
over several steps; the bucketed throughput mode of the earlier rounds is reported beside it as
`throughput_mode_bucketed`), and `strong_scaling` holds BASELINE.json configs[3]: the FIXED 8192-shard
32 x 32 IntersectionCount matrix (+ filter) split over the N ranks, one all-reduce of the 1024-cell partial
matrix per query, the host add beside it, and the same query through the in-library group path
(fbk_group_count_matrix, host / xGMI-peer / RCCL reduce) in `group_api`.  Also in the line: the distribution of
the step time over repeated timed regions, bits-scanned GB/s, the roofline of the dominant kernel
(HIP events around back-to-back launches), the materialising variant, `secondary` = BASELINE.json
configs 3, 4 and 5 at their per-GPU sizes (N = 1), for N > 1 the per-step-collective and
host-add latency modes and the in-library multi-device path (fbk_group_*), and the CPU oracle
timed on this box's host cores as `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

SHARDS_PER_GPU = 1024
REDUCE_BUCKET = 16  # steps per all-reduce of the bucketed throughput mode (reported beside the headline when N > 1)
PER_QUERY_RING = 32  # result cells in rotation when every step carries its own all-reduce (N > 1 headline)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def dist_of(xs):
    """median / p10 / p90 of a list of samples"""
    v = sorted(xs)
    n = len(v)
    return {"median": v[n // 2], "p10": v[n // 10], "p90": v[min(n - 1, (n * 9) // 10)], "n": n}


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1: start the N ranks (one per GPU) and pass rank 0's line through."""
    import torch

    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = dict(os.environ)
    if n_dev < args.gpus:
        # fewer devices than ranks: the ranks share the devices (rank r -> device r mod n_dev) and the
        # collectives run over gloo.  This is the complete N > 1 code path on a smaller box, NOT a
        # scaling measurement; the output line says "oversubscribed": true.
        env["FBK_BENCH_BACKEND"] = "gloo"
        print(f"[bench] {n_dev} device(s) visible for --gpus {args.gpus}: ranks share devices, gloo collectives", file=sys.stderr, flush=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def cpu_baseline(wa: np.ndarray, wb: np.ndarray, budget_s: float = 15.0):
    """The CPU restatement of the Go path (oracle/roaring_oracle.c) on this box's host
    cores: same inputs, one pthread worker per shard chunk (the reference runs one
    goroutine per shard over NumCPU pool workers, executor.go:6723-6737)."""
    import ctypes as C

    src = [os.path.join(ROOT, "oracle", f) for f in ("roaring_oracle.c", "bsi_oracle.c")]
    so = os.path.join(ROOT, "oracle", "libroaring_oracle.so")
    build = "portable gcc -O3 -mpopcnt"
    try:  # a -march=native build for the box we are on (the committed .so is portable)
        tmp = os.path.join(tempfile.mkdtemp(prefix="fbk_orc_"), "liborc_native.so")
        subprocess.check_call(
            ["gcc", "-O3", "-march=native", "-std=gnu99", "-fPIC", "-pthread", "-shared", "-o", tmp] + src, stderr=subprocess.DEVNULL
        )
        so, build = tmp, "gcc -O3 -march=native"
    except Exception:
        pass
    lib = C.CDLL(so)
    f = lib.orc_dense_intersection_count_mt
    f.restype = C.c_uint64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_uint64]
    n = wa.shape[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, n))
    counts = np.zeros(n, dtype=np.uint64)

    def run(threads, passes):
        t0 = time.perf_counter()
        tot = f(wa.ctypes.data, wb.ctypes.data, n, counts.ctypes.data, threads, passes)
        return time.perf_counter() - t0, tot

    def timed(threads, target_s):
        """Grow the pass count until one call lasts >= target_s (thread start-up and the
        first touch of the pages are then negligible); returns (seconds, passes, total)."""
        passes = 1
        while True:
            t, tot = run(threads, passes)
            if t >= target_s or passes >= 1 << 20:
                return t, passes, tot
            passes = max(passes * 2, int(passes * 1.2 * target_s / max(t, 1e-4)))

    t1, p1, tot = timed(1, budget_s / 4)
    single = n * 16 * p1 / t1
    tm, pm, tot = timed(cores, budget_s / 2)
    # the same source without auto-vectorisation: the closer stand-in for Go's scalar codegen of
    # the 1024-word AND + POPCNT loop (roaring.go:1088-1090)
    novec = None
    try:
        tmp2 = os.path.join(tempfile.mkdtemp(prefix="fbk_orc_"), "liborc_novec.so")
        subprocess.check_call(
            ["gcc", "-O3", "-march=native", "-fno-tree-vectorize", "-fno-tree-slp-vectorize", "-std=gnu99", "-fPIC", "-pthread", "-shared", "-o", tmp2]
            + src,
            stderr=subprocess.DEVNULL,
        )
        f2 = C.CDLL(tmp2).orc_dense_intersection_count_mt
        f2.restype = C.c_uint64
        f2.argtypes = f.argtypes
        f_vec, f = f, f2
        tv, pv, totv = timed(1, budget_s / 4)
        f = f_vec
        novec = n * 16 * pv / tv
    except Exception:
        pass
    # the shape the reference's executor sees: every container its own heap object behind a key-sorted slice (a Bitmap per
    # row), all 1024 shards walked once per pass by the pool of threads — 256 MiB streamed, nothing cache-resident by design
    streaming = None
    try:
        from oracle import pybatch as PB

        OA, OB = PB.RowSet.from_dense(wa), PB.RowSet.from_dense(wb)
        idx = np.arange(n)
        got = PB.intersection_count(OA, idx, OB, idx, nthreads=cores)
        assert int(got.sum()) == int(tot)
        reps = 8
        while True:  # one pool of threads for all passes of a call (thread start-up is not what is timed)
            t0 = time.perf_counter()
            got = PB.intersection_count_repeat(OA, idx, OB, idx, reps, nthreads=cores)
            ts = (time.perf_counter() - t0) / reps
            if ts * reps >= budget_s / 5 or reps >= 1 << 16:
                break
            reps *= 4
        assert int(got.sum()) == int(tot)
        streaming = {"value": n * 16 / ts, "unit": "set-ops/s", "cores": cores, "bits_scanned_GBps": 2 * n * 16 * 8192 / ts / 1e9,
                     "sample": f"Bitmap.IntersectionCount over {n} shard row pairs held as the oracle's Bitmaps (one heap object per container), "
                               f"one pass = every shard once, {reps} passes on {cores} threads (oracle/batch_oracle.c)"}
        OA.free()
        OB.free()
    except Exception as e:  # noqa: BLE001
        streaming = {"error": str(e)}
    return {
        "value": n * 16 * pm / tm,
        "unit": "set-ops/s",
        "cores": cores,
        "streaming_pass": streaming,
        "kind": "port",
        "sample": f"full workload ({n} shards x 2 rows, 256 MiB) x {pm} passes on {cores} threads ({tm:.1f} s), "
        f"C restatement of the Go path (oracle/roaring_oracle.c) built with {build}; each thread re-scans its own "
        f"{max(1, n // cores)}-shard chunk ({max(1, n // cores) * 256} KiB), i.e. the CPU figure is cache-resident: an upper bound for the CPU",
        "bits_scanned_GBps": 2 * n * 16 * 8192 * pm / tm / 1e9,
        "single_thread_set_ops_per_s": single,
        "single_thread_no_autovectorize_set_ops_per_s": novec,
        "total_count": int(tot),
    }


# ---- secondary configurations (N = 1): BASELINE.json configs 3, 4, 5 ------------------------------


def _cpu_time(fn, min_s=1.0):
    """seconds per call of fn on one host thread (repeated until min_s has elapsed)"""
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_s:
            return dt / n


def _timed_call(torch, stream, fn, iters, warm=2, ctx=None):
    """One C-ABI call end to end: GPU time between the first and the last operation it enqueues (HIP
    events on the library's stream) and host wall time, over `iters` calls; with `ctx`, also the
    duration of the call's dominant kernel (library option time_kernels: HIP events recorded on the
    library's stream right before and after that launch)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gpu, wall, kern = [], [], []
    if ctx is not None:
        ctx.set_option("time_kernels", 1)
    try:
        for _ in range(iters):
            e0.record(stream)
            t0 = time.perf_counter()
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) * 1e6)
            gpu.append(e0.elapsed_time(e1) * 1e3)
            if ctx is not None:
                kern.append(ctx.get_option("last_kernel_ns") / 1e3)
    finally:
        if ctx is not None:
            ctx.set_option("time_kernels", 0)
    if ctx is not None:
        return dist_of(gpu), dist_of(wall), dist_of(kern)
    return dist_of(gpu), dist_of(wall)


def _entry(name, kernel, alg_bytes, gpu_us, wall_us, kernel_us=None, **extra):
    t = gpu_us["median"] * 1e-6
    ident, _, name = name.rpartition("|")  # "short id|descriptive name": the id is what the compact stdout line carries (bench_line.py)
    out = {
        "id": ident or None,
        "name": name,
        "kernel": kernel,
        "algorithmic_bytes": int(alg_bytes),
        "gpu_us": gpu_us,
        "wall_us": wall_us,
        "achieved_GBps": alg_bytes / t / 1e9,
        "frac": alg_bytes / t / 1e9 / HBM_PEAK_GBPS,
        "timing": "gpu_us / wall_us / achieved_GBps / frac: one C-ABI call end to end (index upload, memsets, launches, reduce, result download), "
                  "HIP events on the stream; kernel_us / kernel_GBps / kernel_frac: the named kernel(s) alone, HIP events recorded by the library "
                  "around the launch (option time_kernels); rocprofv3 summaries of the same kernels are in profiles/",
    }
    if kernel_us is not None:
        out["kernel_us"] = kernel_us
        out["kernel_GBps"] = alg_bytes / (kernel_us["median"] * 1e-6) / 1e9
        out["kernel_frac"] = out["kernel_GBps"] / HBM_PEAK_GBPS
    out.update(extra)
    return out


def gpu_random_rows(torch, dev, n_rows: int, seed: int) -> np.ndarray:
    """n_rows x 16 x 1024 uint64 of fair random bits, generated on the device (numpy needs ~1 s per GB)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.randint(-(2**63), 2**63 - 1, (n_rows, 16, 1024), dtype=torch.int64, device=dev, generator=g)
    return t.cpu().numpy().view(np.uint64)


def _cpu_batch_time(fn, min_s=0.5):
    """seconds per call of a batch-oracle call (all shards on all host threads), repeated until min_s has elapsed"""
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_s:
            return dt / n


def _timed_query(torch, stream, q, iters, ctx, warm=2):
    """A prepared query (fbk_query_*): GPU time of ONE execution — memset + kernel(s) + reduce, nothing else is
    enqueued — between HIP events on the library's stream, and the dominant kernel alone (option time_kernels)."""
    return _timed_call(torch, stream, q.run, iters, warm=warm, ctx=ctx)


def secondary_configs(torch, dev, ctx, stream, pre3, args, want_cpu):
    """BASELINE.json configs 3, 4, 5 at their per-GPU sizes.  Timed through prepared queries (row lists and result
    buffers resident, an execution is launch-only); with the CPU leg EVERY shard of every configuration is compared
    with the oracle (oracle/batch_oracle.c: the restated reference calls over all shards on the host threads)."""
    from featurebase_amd import lib as L

    PB = None
    if want_cpu:
        from oracle import pybatch as PB
    out = []
    iters = args.secondary_iters
    timing_note = ("gpu_us / wall_us: ONE execution of the prepared query (fbk_query_run: memset + kernel(s) + reduce over shards; row lists and "
                   "result buffers resident), HIP events on the library's stream; kernel_us: the named kernel alone (option time_kernels); "
                   "call_us: the one-shot C-ABI call end to end (index upload, launches, result download, synchronisation)")
    # ---- config 3: mixed containers, Union-of-64 then IntersectionCount; TopK shape; GroupBy 32 x 32 ----
    if pre3 is not None:
        rows, groups, filt, gen_s = pre3
        n3 = groups.shape[0]
        # (the flattened form — descriptor table + ONE payload buffer — is what a caller hands to fbk_batch_upload; building it
        # from the generator's 266 k numpy pieces is not part of the upload)
        d3, p3, fd3, fp3 = rows.descs(), rows.payload(), filt.descs(), filt.payload()
        t0 = time.perf_counter()
        batch = ctx.upload_flat(d3, p3, rows.n_rows)
        F = ctx.upload_flat(fd3, fp3, filt.n_rows)
        up_s = time.perf_counter() - t0
        fidx = np.arange(n3)
        nbytes = rows.bytes + filt.bytes
        ncont = len(rows.key) + len(filt.key)
        q_fold = ctx.prepare_fold_intersection_count(L.OP_OR, batch, groups, F, fidx)
        q_top = ctx.prepare_count_matrix(batch, groups, F, fidx.reshape(-1, 1), keep_per_shard=True)
        q_gb = ctx.prepare_count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, fidx, keep_per_shard=True)
        for q in (q_fold, q_top, q_gb):
            q.run()
        got3, topn3, mat3 = q_fold.read(), q_top.read(per_shard=True)[1], q_gb.read(per_shard=True)[1]
        assert (got3 == ctx.union_n_intersection_count(batch, groups, F, fidx)).all(), "config 3: prepared and one-shot fold disagree"
        cpu3 = None
        if want_cpu:
            OA, OF = PB.RowSet.from_flat(d3, p3, rows.n_rows), PB.RowSet.from_flat(fd3, fp3, filt.n_rows)
            e_fold, _ = PB.union_n_intersection_count(OA, groups, OF, fidx)
            assert (got3 == e_fold).all(), "config 3: GPU and oracle disagree (Union-of-64 then IntersectionCount)"
            assert (topn3[:, :, 0] == PB.topk_counts(OA, groups, OF, fidx)).all(), "config 3 TopN: GPU and oracle disagree"
            assert (mat3 == PB.count_matrix(OA, groups[:, :32], OA, groups[:, 32:], OF, fidx)).all(), "config 3 GroupBy: GPU and oracle disagree"
            t_cpu = _cpu_batch_time(lambda: PB.union_n_intersection_count(OA, groups, OF, fidx))
            t_cpu1 = _cpu_batch_time(lambda: PB.union_n_intersection_count(OA, groups[:8], OF, fidx[:8], nthreads=1)) / 8
            t_cpu_gb = _cpu_batch_time(lambda: PB.count_matrix(OA, groups[:, :32], OA, groups[:, 32:], OF, fidx))
            cpu3 = {"kind": "port", "cores": PB.threads(), "sample": f"all {n3} shards (64 mixed rows + filter each), oracle Bitmap.Union(63 others) + IntersectionCount, one shard per host thread",
                    "value": n3 * 16 * 64 / t_cpu, "unit": "set-ops/s", "all_shards_s": t_cpu, "per_shard_one_thread_s": t_cpu1, "groupby_32x32_all_shards_s": t_cpu_gb}
        common = {"shards": n3, "containers": ncont, "host_gen_s": gen_s, "upload_s": up_s, "upload_GBps": nbytes / up_s / 1e9, "timing": timing_note,
                  "parity": f"every one of the {n3} shards bit-exact against the oracle" if want_cpu else "unchecked (--no-cpu-baseline)"}

        def call_us(fn, n=max(5, iters // 2)):
            return _timed_call(torch, stream, fn, n)[0]

        g, w, kq = _timed_query(torch, stream, q_fold, iters, ctx)
        out.append(_entry("c3.union64_icount|config3: Union-of-64 rows then IntersectionCount(filter), fused, mixed array/run/bitmap rows (rank-law density 0.001-0.5)",
                          "k_fold_scatter<OR>", nbytes + 8 * n3, g, w, kq, set_ops_per_s=n3 * 16 * 64 / (g["median"] * 1e-6), cpu_baseline=cpu3,
                          call_us=call_us(lambda: ctx.union_n_intersection_count(batch, groups, F, fidx)), **common))
        g, w, kq = _timed_query(torch, stream, q_top, iters, ctx)
        out.append(_entry("c3.rows_vs_filter|config3 rows, TopN/TopK shape: 64 rows x 1 filter row per shard", "k_rows_vs_filter", nbytes + 8 * 64 * n3, g, w, kq,
                          set_ops_per_s=n3 * 16 * 64 / (g["median"] * 1e-6), call_us=call_us(lambda: ctx.count_matrix(batch, groups, F, fidx.reshape(-1, 1))), **common))
        g, w, kq = _timed_query(torch, stream, q_gb, max(5, iters // 2), ctx)
        # heavy containers (run containers, arrays of more than 2048 values) are read through dense shadows the library builds per
        # batch on the first count matrix (option matrix_shadow, fbk.hip heavy_shadow): what that costs in memory and traffic
        dd = d3
        heavy = int((((dd["type"] == 3) & (dd["n"] != 0)) | ((dd["type"] == 1) & (dd["len"] > 2048))).sum())
        heavy_payload = int((dd["len"][(dd["type"] == 3) & (dd["n"] != 0)].astype(np.int64) * 4).sum() + (dd["len"][(dd["type"] == 1) & (dd["len"] > 2048)].astype(np.int64) * 2).sum())
        out.append(_entry("c3.groupby32x32|config3 rows, GroupBy 32 x 32 (+ filter) on mixed rows: decode inside the matrix-core kernel, heavy containers through dense shadows (default)", "k_count_matrix_fusedq",
                          nbytes + 8 * 1024 * n3, g, w, kq, set_ops_per_s=n3 * 16 * 1024 / (g["median"] * 1e-6),
                          call_us=call_us(lambda: ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, fidx)),
                          heavy_row_shadows={"containers": heavy, "resident_bytes": heavy * 8192 + 16 * 16 * rows.n_rows, "built": "once per batch, on the first count matrix that reads it",
                                             "bytes_read_per_query": nbytes - heavy_payload + heavy * 8192,
                                             "note": "frac / kernel_frac are quoted on the ENCODED bytes (the algorithmic bytes of the query); the kernel itself reads bytes_read_per_query"},
                          hbm_note="round 5: the kernel runs a prepared program (k_fused_program) with producer waves specialised on array / bitmap rows and loads two stages ahead; with the heavy rows shadowed the bytes it reads (bytes_read_per_query) take ~180 us at the achievable HBM rate, the consumers alone ~180 us (scripts/fused_ablate.py; DESIGN.md section 9)", **common))
        # row pairs of the same rows (RowSegment.IntersectionCount / Intersect on non-dense rows): rows 0..31 against rows 32..63 of every shard
        pa, pb = groups[:, :32].reshape(-1), groups[:, 32:].reshape(-1)
        plan = ctx.plan(batch, pa, batch, pb)
        plan.intersection_count()
        pc = plan.read()
        if want_cpu:
            assert (pc == PB.intersection_count(OA, pa, OA, pb)).all(), "config 3 row pairs: GPU and oracle disagree"
        g, w = _timed_call(torch, stream, plan.intersection_count, iters)
        out.append(_entry(f"c3.pairs_icount|config3 rows, {pa.size} row pairs (rows 0..31 x rows 32..63 of every shard): IntersectionCount, launch-only plan", "k_icount2", rows.bytes, g, w,
                          set_ops_per_s=pa.size * 16 / (g["median"] * 1e-6), **common))
        g, w = _timed_call(torch, stream, lambda: plan.setop(L.OP_AND), iters)
        out.append(_entry(f"c3.pairs_intersect_cells|config3 rows, {pa.size} row pairs: Intersect materialised (8 KiB cells), launch-only plan", "k_setop2<AND>", rows.bytes + pa.size * 16 * 8192, g, w,
                          set_ops_per_s=pa.size * 16 / (g["median"] * 1e-6), **common))
        # ... and with Container.optimize() applied inside the set-op kernel (round 4): the encoded containers are the only bytes written
        plan.setop(L.OP_AND, L.SETOP_OPTIMIZE)
        so_counts = plan.read()
        so_bytes = plan.output().info()[2]
        if want_cpu:
            eo, eo_cnt = PB.setop(PB.OP_AND, OA, pa, OA, pb)
            assert (so_counts == eo_cnt).all(), "config 3 row pairs, Intersect + optimize: cardinalities differ from the oracle"
            ds, ps_, nrs = plan.output().download_flat()
            assert (PB.RowSet.from_flat(ds, ps_, nrs).words() == eo.words()).all(), "config 3 row pairs, Intersect + optimize: bit content differs from the oracle"
            eo.free()
        g, w = _timed_call(torch, stream, lambda: plan.setop(L.OP_AND, L.SETOP_OPTIMIZE), iters)
        # for scale: the one-shot call with optimize() inside the kernel, and with the round-2/3 pipeline (bitmap / small-array cells,
        # then the separate re-encode pass: plan, two scans, a host round trip for the arena size, write)
        c_in = call_us(lambda: ctx.setop(L.OP_AND, batch, pa, batch, pb, L.SETOP_OPTIMIZE)[0].free())
        ctx.set_option("setop_direct_encode", 1)
        c_sep = call_us(lambda: ctx.setop(L.OP_AND, batch, pa, batch, pb, L.SETOP_OPTIMIZE)[0].free())
        ctx.set_option("setop_direct_encode", 2)
        out.append(_entry(f"c3.pairs_intersect_optimize|config3 rows, {pa.size} row pairs: Intersect materialised + optimize() inside the kernel (only the encoded containers are written), launch-only plan", "k_setop2<AND> (optimize)",
                          rows.bytes + so_bytes, g, w, set_ops_per_s=pa.size * 16 / (g["median"] * 1e-6), output_payload_bytes=so_bytes, call_us=c_in,
                          call_us_with_the_separate_reencode_pass=c_sep, **common))
        plan.free()
        # Union-of-64 MATERIALISED + optimize(): the prepared query (group lists and the output batch resident: memset + one launch of the
        # fold kernel, which encodes in its epilogue) beside the one-shot call; every result container compared with the oracle's
        # union re-encoded by optimize() — encoding, cardinality and payload bytes
        q_un = ctx.prepare_fold(L.OP_OR, batch, groups, L.SETOP_OPTIMIZE)
        q_un.run()
        un_counts = q_un.read()
        un_bytes = q_un.output().info()[2]
        if want_cpu:
            eu, eu_cnt = PB.union_n(OA, groups)
            assert (un_counts == eu_cnt).all(), "config 3 materialised union: cardinalities differ from the oracle"
            du, pu, nru = q_un.output().download_flat()
            assert (PB.RowSet.from_flat(du, pu, nru).words() == eu.words()).all(), "config 3 materialised union: bit content differs from the oracle"
            eu.free()
        g, w, kq = _timed_query(torch, stream, q_un, iters, ctx)
        out.append(_entry("c3.union64_optimize|config3 rows, Union-of-64 materialised + optimize(): prepared query, Container.optimize() in the fold kernel's epilogue", "k_fold_scatter<OR, optimize>",
                          nbytes + un_bytes, g, w, kq, output_payload_bytes=un_bytes,
                          call_us=call_us(lambda: ctx.union_n(batch, groups, L.SETOP_OPTIMIZE)[0].free()), **common))
        q_un.free()
        # TopN with its ordering on the device (radix sort of the 64 totals; the count matrix entry above stops at the per-shard counts)
        q_tn = ctx.prepare_topn(batch, groups, 10, F, fidx)
        q_tn.run()
        tn_idx, tn_cnt = q_tn.read()
        if want_cpu:
            tot_e = PB.topk_counts(OA, groups, OF, fidx).sum(axis=0)
            order = sorted([i for i in range(64) if tot_e[i]], key=lambda i: (-int(tot_e[i]), i))[:10]
            assert tn_idx.tolist() == order and [int(c) for c in tn_cnt] == [int(tot_e[i]) for i in order], "config 3 TopN: GPU and oracle disagree"
        g, w, kq = _timed_query(torch, stream, q_tn, iters, ctx)
        out.append(_entry("c3.topn10|config3 rows, TopN(n = 10) of 64 rows against the filter row: counts, sum over shards and ordering on the device, prepared query", "k_rows_vs_filter", nbytes + 8 * 64 * n3,
                          g, w, kq, call_us=call_us(lambda: ctx.topn(batch, groups, 10, F, fidx)), **common))
        q_tn.free()
        for q in (q_fold, q_top, q_gb):
            q.free()
        batch.free()
        F.free()
        if want_cpu:
            OA.free()
            OF.free()
    # ---- config 4 (per-GPU slice of the 8192-shard configuration): 32 x 32 count matrix + filter, dense ----
    n4, n_a, n_b = args.shards4, 32, 32
    if n4:
        t0 = time.perf_counter()
        wa, wb, wf = gpu_random_rows(torch, dev, n4 * n_a, 41), gpu_random_rows(torch, dev, n4 * n_b, 42), gpu_random_rows(torch, dev, n4, 43)
        gen_s = time.perf_counter() - t0
        A, B, F = ctx.upload_dense(wa), ctx.upload_dense(wb), ctx.upload_dense(wf)
        ra, rb, rf = np.arange(n4 * n_a).reshape(n4, n_a), np.arange(n4 * n_b).reshape(n4, n_b), np.arange(n4)
        q4 = ctx.prepare_count_matrix(A, ra, B, rb, F, rf, keep_per_shard=True)
        q4.run()
        tot, ps4 = q4.read(per_shard=True)
        exp = int(sum(np.bitwise_count(wa[s * n_a + 3] & wb[s * n_b + 5] & wf[s]).sum() for s in range(n4)))
        assert int(tot[3, 5]) == exp, "config 4: GPU and numpy disagree"
        cpu4 = None
        if want_cpu:
            OA, OB, OF = PB.RowSet.from_dense(wa), PB.RowSet.from_dense(wb), PB.RowSet.from_dense(wf)
            t0 = time.perf_counter()
            e4 = PB.count_matrix(OA, ra, OB, rb, OF, rf)
            t_cpu = time.perf_counter() - t0
            assert (ps4 == e4).all() and (tot == e4.sum(axis=0)).all(), "config 4: GPU and oracle disagree"
            t_cpu1 = _cpu_batch_time(lambda: PB.count_matrix(OA, ra[:2], OB, rb[:2], OF, rf[:2], nthreads=1)) / 2
            cpu4 = {"kind": "port", "cores": PB.threads(), "sample": f"all {n4} shards (32 x 32 dense rows + filter), oracle groupByIterator counts, one shard per host thread",
                    "value": n4 * 16 * n_a * n_b / t_cpu, "unit": "set-ops/s", "all_shards_s": t_cpu, "per_shard_one_thread_s": t_cpu1}
            for o in (OA, OB, OF):
                o.free()
        nbytes = n4 * (n_a + n_b + 1) * 16 * 8192
        # (as for the log-uniform slice below: the first launches after the rows were written run 4-15 % slower than the sustained
        # rate; the entry's kernel_us is the SUSTAINED one — SURVEY 8d's protocol is "warm-up, then >= 20 timed" — the first five
        # launches are reported beside it)
        _, _, kq4_first = _timed_query(torch, stream, q4, 5, ctx, warm=0)
        g, w, kq = _timed_query(torch, stream, q4, 20, ctx, warm=10)
        out.append(_entry(f"c4.dense_slice|config4 slice: {n4} shards x (32 x 32 rows + filter), dense bitmaps, IntersectionCount matrix", "k_count_matrix_mfma",
                          nbytes + 8 * n_a * n_b * n4, g, w, kq, shards=n4, host_gen_s=gen_s, set_ops_per_s=n4 * 16 * n_a * n_b / (g["median"] * 1e-6),
                          pair_bits_scanned_GBps=n4 * n_a * n_b * 2 * 16 * 8192 / (g["median"] * 1e-6) / 1e9, cpu_baseline=cpu4, timing=timing_note,
                          kernel_us_first_launches=kq4_first,
                          call_us=_timed_call(torch, stream, lambda: ctx.count_matrix(A, ra, B, rb, F, rf), 5)[0],
                          parity=f"every one of the {n4} per-shard matrices bit-exact against the oracle" if want_cpu else "cell (3,5) vs numpy over all shards"))
        q4.free()
        for b in (A, B, F):
            b.free()
        del wa, wb, wf
    # ---- config 4 on the input SURVEY.md 8d specifies: densities LOG-UNIFORM in [0.001, 0.5] (2/3 of the rows are array rows) ----
    n4m = args.shards4_mixed
    if n4m:
        import datagen as D

        t0 = time.perf_counter()
        d4, p4, nr4, g4, fd4, fp4, nbytes = D.config3_flat_subprocess(n4m, n_a + n_b, 4000, config4=True)
        gen_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        batch4, F4 = ctx.upload_flat(d4, p4, nr4), ctx.upload_flat(fd4, fp4, n4m)
        up_s = time.perf_counter() - t0
        ga, gb, fidx = np.ascontiguousarray(g4[:, :n_a]), np.ascontiguousarray(g4[:, n_a:]), np.arange(n4m)
        q4m = ctx.prepare_count_matrix(batch4, ga, batch4, gb, F4, fidx, keep_per_shard=True)
        q4m.run()
        tot, ps4 = q4m.read(per_shard=True)
        cpu4m = None
        if want_cpu:
            OA, OF = PB.RowSet.from_flat(d4, p4, nr4), PB.RowSet.from_flat(fd4, fp4, n4m)
            t0 = time.perf_counter()
            e4 = PB.count_matrix(OA, ga, OA, gb, OF, fidx)
            t_cpu = time.perf_counter() - t0
            assert (ps4 == e4).all() and (tot == e4.sum(axis=0)).all(), "config 4 (log-uniform rows): GPU and oracle disagree"
            cpu4m = {"kind": "port", "cores": PB.threads(), "sample": f"all {n4m} shards (32 x 32 log-uniform rows + filter), oracle groupByIterator counts, one shard per host thread",
                     "value": n4m * 16 * n_a * n_b / t_cpu, "unit": "set-ops/s", "all_shards_s": t_cpu}
            OA.free()
            OF.free()
        types = np.bincount(d4["type"], minlength=4)
        # (this kernel's first ~10 launches after ANY pause of the device — here: seconds of host-only oracle work — run up to 30 %
        # slower than its sustained rate: the shader clock dips to ~1.7 GHz on the load step and is back at 2.2 GHz after ~15 ms,
        # scripts/first_launches.py, profiles/r06_first_launches.txt.  Both are reported; the entry's kernel_us is the SUSTAINED one,
        # SURVEY 8d's protocol being "warm-up, then >= 20 timed")
        _, _, kq_first = _timed_query(torch, stream, q4m, 5, ctx, warm=0)
        g, w, kq = _timed_query(torch, stream, q4m, 20, ctx, warm=10)
        out.append(_entry(f"c4.loguniform_slice|config4 slice as SURVEY 8d writes it: {n4m} shards x (32 x 32 rows, densities log-uniform [0.001, 0.5] + filter p = 0.5), IntersectionCount matrix on ENCODED rows",
                          "k_count_matrix_fusedq", nbytes + 8 * n_a * n_b * n4m, g, w, kq, shards=n4m, host_gen_s=gen_s, upload_s=up_s, set_ops_per_s=n4m * 16 * n_a * n_b / (g["median"] * 1e-6),
                          containers={"array": int(types[1]), "bitmap": int(types[2]), "run": int(types[3])}, cpu_baseline=cpu4m, timing=timing_note,
                          kernel_us_first_launches=kq_first,
                          note="frac is quoted on the ENCODED bytes (arrays 2 n, bitmaps 8192: the algorithmic bytes of SURVEY 8d); the dense-only slice above reads 2.1 x these bytes per shard",
                          parity=f"every one of the {n4m} per-shard matrices bit-exact against the oracle" if want_cpu else "unchecked (--no-cpu-baseline)"))
        q4m.free()
        for b in (batch4, F4):
            b.free()
        del d4, p4
    # ---- config 5: BSI Range(> k) + Sum, 64 bit planes + exists + sign, 100 M columns = 96 shards ----
    n5, depth = args.shards5, 64
    if n5:
        w = gpu_random_rows(torch, dev, n5 * (depth + 2), 51).reshape(n5, depth + 2, 16, 1024)
        w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)  # exists: every column has a value
        w[-1, 0, 6:] = 0  # last shard partial (100M columns = 95 full shards + 385 280 columns)
        w[-1, 0, 5, 900:] = 0
        w[:, 1:] &= w[:, :1]  # planes only where a value exists
        batch = ctx.upload_dense(w.reshape(-1))
        base = np.arange(n5, dtype=np.uint32) * (depth + 2)
        idx5 = np.arange(n5)
        kk = 1 << 62
        plane_bytes = n5 * 16 * 8192
        rng_out, rng_cnt = ctx.bsi_range(batch, base, L.BSI_GT, depth, kk)
        q_sum = ctx.prepare_bsi_sum(batch, base, depth, filt=rng_out, rows_f=idx5)
        q_fused = ctx.prepare_bsi_sum(batch, base, depth, L.BSI_GT, kk)
        q_sum.run()
        q_fused.run()
        sums, cnts = q_sum.read()
        fsum, fcnt = q_fused.read()
        assert (fsum == sums).all() and (fcnt == cnts).all(), "config 5 fused Range+Sum: differs from Range then Sum"
        cpu5 = None
        if want_cpu:
            OA = PB.RowSet.from_dense(w.reshape(-1, 16, 1024))
            e_rng, e_cnt = PB.bsi_range(OA, base, depth, PB.GT, kk)
            assert (rng_cnt == e_cnt).all(), "config 5 Range: GPU and oracle disagree"
            d5, p5, nr5 = rng_out.download_flat()
            assert (PB.RowSet.from_flat(d5, p5, nr5).words() == e_rng.words()).all(), "config 5 Range: bit content differs from the oracle"
            e_sum, e_c = PB.bsi_sum(OA, base, depth, e_rng, idx5)
            assert (sums == e_sum).all() and (cnts == e_c).all(), "config 5 Sum: GPU and oracle disagree"
            t_rng = _cpu_batch_time(lambda: PB.bsi_range(OA, base, depth, PB.GT, kk)[0].free())
            t_sum = _cpu_batch_time(lambda: PB.bsi_sum(OA, base, depth, e_rng, idx5))
            cpu5 = {"kind": "port", "cores": PB.threads(), "sample": f"all {n5} shards (66 dense rows each), oracle fragment.rangeOp(GT) / fragment.sum, one shard per host thread",
                    "range_all_shards_s": t_rng, "sum_all_shards_s": t_sum}
            e_rng.free()
            OA.free()
        par5 = f"every one of the {n5} shards bit-exact against the oracle" if want_cpu else "unchecked"
        q_rng = ctx.prepare_bsi_range(batch, base, L.BSI_GT, depth, kk)
        q_rng.run()
        assert (q_rng.read() == rng_cnt).all(), "config 5: prepared Range differs from the one-shot call"
        g, wl, kq = _timed_query(torch, stream, q_rng, iters, ctx)
        out.append(_entry(f"c5.range|config5: BSI Range(> 2^62), {n5} shards x (64 planes + exists + sign), dense: prepared query, the result rows stay on the device", "k_bsi_range_slot",
                          plane_bytes * (depth + 3), g, wl, kq, shards=n5, cpu_baseline=cpu5, parity=par5, timing=timing_note,
                          call_us=_timed_call(torch, stream, lambda: ctx.bsi_range(batch, base, L.BSI_GT, depth, kk)[0].free(), 5)[0]))
        q_rng.free()
        g, wl, kq = _timed_query(torch, stream, q_sum, iters, ctx)
        out.append(_entry("c5.sum|config5: BSI Sum(filter = the Range result)", "k_bsi_sum_slot", plane_bytes * (depth + 3), g, wl, kq, shards=n5, parity=par5, timing=timing_note,
                          call_us=_timed_call(torch, stream, lambda: ctx.bsi_sum(batch, base, depth, rng_out, idx5), 5)[0]))
        g, wl, kq = _timed_query(torch, stream, q_fused, iters, ctx)
        out.append(_entry("c5.range_sum_fused|config5 fused: Sum(Range(> 2^62)) of the same field, one pass over the planes", "k_bsi_range_sum_half", plane_bytes * (depth + 2), g, wl, kq, shards=n5,
                          timing=timing_note, call_us=_timed_call(torch, stream, lambda: ctx.bsi_range_sum(batch, base, L.BSI_GT, depth, kk), 5)[0],
                          parity="equal to the Range-then-Sum totals of the entry above (oracle-checked), every shard"))
        q_sum.free()
        q_fused.free()
        rng_out.free()
        batch.free()
    return out


def config4_strong(torch, dist, fdist, dev, ctx, stream, rank, world, args, cpu_group, want_cpu):
    """BASELINE.json configs[3]: the FIXED problem of 8192 shards x (32 x 32 rows + filter row), split over the
    ranks (shard s on rank s mod N: executor.go:6449-6533), GroupBy-style IntersectionCount matrix per shard
    (groupByIterator, executor.go:8880-8934), the 1024-cell partial matrices of the ranks summed by ONE all-reduce
    per query (mergeGroupCounts across nodes, executor.go:3728-3762).  Rows are generated on the device (70 GB at
    N = 1).  Returns this rank's dict; times are this rank's clock (the caller takes the max over ranks)."""
    total, n_a, n_b = args.shards4_total, 32, 32
    mine = fdist.shards_for_rank(total, rank, world)
    ns = len(mine)
    t0 = time.perf_counter()

    def gen(n_rows, seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return torch.randint(-(2**63), 2**63 - 1, (n_rows, 16, 1024), dtype=torch.int64, device=dev, generator=g)

    ta, tb, tf = gen(ns * n_a, 4100 + 3 * rank), gen(ns * n_b, 4101 + 3 * rank), gen(ns, 4102 + 3 * rank)
    torch.cuda.synchronize()
    A, B, F = ctx.upload_dense_device(ta.data_ptr(), ns * n_a), ctx.upload_dense_device(tb.data_ptr(), ns * n_b), ctx.upload_dense_device(tf.data_ptr(), ns)
    resident_s = time.perf_counter() - t0
    ra, rb, rf = np.arange(ns * n_a).reshape(ns, n_a), np.arange(ns * n_b).reshape(ns, n_b), np.arange(ns)
    q = ctx.prepare_count_matrix(A, ra, B, rb, F, rf, keep_per_shard=True)
    q.run()
    local, ps = q.read(per_shard=True)
    parity = "unchecked (--no-cpu-baseline)"
    if want_cpu:
        # EVERY shard of this rank against the oracle, a chunk of shards at a time (the rank's rows are up to 70 GB: they
        # stay on the device, the generator's tensors are copied to the host chunk by chunk for the checker)
        from oracle import pybatch as PB

        t_chk = time.perf_counter()
        chunk = 256
        for c0 in range(0, ns, chunk):
            c1 = min(ns, c0 + chunk)
            OA, OB, OF = (PB.RowSet.from_dense(t[c0 * k: c1 * k].cpu().numpy().view(np.uint64)) for t, k in ((ta, n_a), (tb, n_b), (tf, 1)))
            m = c1 - c0
            e = PB.count_matrix(OA, np.arange(m * n_a).reshape(m, n_a), OB, np.arange(m * n_b).reshape(m, n_b), OF, np.arange(m))
            assert (ps[c0:c1] == e).all(), f"config 4 strong: GPU and oracle disagree in shards {c0}..{c1 - 1} of rank {rank}"
            for o in (OA, OB, OF):
                o.free()
        parity = f"every one of this rank's {ns} shards bit-exact against the oracle ({time.perf_counter() - t_chk:.1f} s, {PB.threads()} host threads)"
    del ta, tb, tf
    torch.cuda.empty_cache()
    assert (ps.sum(axis=0) == local).all()
    expected = torch.from_numpy(local.view(np.int64).reshape(-1).copy()).to(dev)
    if world > 1:
        dist.all_reduce(expected)
    expected = expected.cpu().numpy().view(np.uint64)
    # the kernel alone (HIP events by the library around the launch) and one execution of the prepared query
    g_us, w_us, k_us = _timed_query(torch, stream, q, 5, ctx)

    def run_local(cell):
        q.run(cell.data_ptr())

    res = fdist.strong_scaling_queries(run_local, n_a * n_b, args.queries4, dev, expected=expected, sync=torch.cuda.synchronize, depth=4, cpu_group=cpu_group)
    nbytes = ns * (n_a + n_b + 1) * 16 * 8192
    out = {"shards_total": total, "shards_this_rank": ns, "rows": f"{n_a} x {n_b} + filter row per shard, dense bitmaps (generated on the device)", "resident_bytes_this_rank": nbytes,
           "make_resident_s": resident_s, "kernel_us": k_us, "query_gpu_us": g_us, "kernel_GBps_this_rank": nbytes / (k_us["median"] * 1e-6) / 1e9,
           "kernel_frac_of_8TBps": nbytes / (k_us["median"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, "parity": parity, **res}
    q.free()
    for b in (A, B, F):
        b.free()
    torch.cuda.empty_cache()
    out["variants"] = [{"id": "dense", "rows": out["rows"], "ms_per_query_pipelined": res["pipelined_s_per_query"] * 1e3, "kernel": "k_count_matrix_mfma", "kernel_us": k_us["median"],
                        "kernel_frac": out["kernel_frac_of_8TBps"], "parity": parity, "shards_this_rank": ns}]
    if args.shards4_mixed_total:
        out["variants"].append(config4_strong_mixed(torch, dist, fdist, dev, ctx, stream, rank, world, args, cpu_group, want_cpu))
    return out


def config4_strong_mixed(torch, dist, fdist, dev, ctx, stream, rank, world, args, cpu_group, want_cpu):
    """configs[3] strong-scaled on the input SURVEY.md 8d specifies (fields A and B: 32 rows each, densities log-uniform in
    [0.001, 0.5]: two thirds array rows; filter p = 0.5; encodings by optimize()).  The rank owns a contiguous block of
    shards (SURVEY 8e allows either dealing; the rows are i.i.d.), generated on the host in slices of <= 1024 shards — one
    batch and one prepared query per slice, a "query" runs them back to back accumulating into one 1024-cell matrix — so
    that the host never holds more than one slice (4.3 GB) of the 34 GB an N = 1 run makes resident."""
    import datagen as D

    total, n_a, n_b = args.shards4_mixed_total, 32, 32
    per = [total // world + (1 if r < total % world else 0) for r in range(world)]
    first, ns = sum(per[:rank]), per[rank]
    t0 = time.perf_counter()
    qs, keep, nbytes, types = [], [], 0, np.zeros(4, dtype=np.int64)
    local = np.zeros((n_a, n_b), dtype=np.uint64)
    t_chk, PB = 0.0, None
    if want_cpu:
        from oracle import pybatch as PB
    first_run_us = []
    for s0 in range(0, ns, 1024):
        m = min(1024, ns - s0)
        d, p, nr, g, fd, fp, nb = D.config3_flat_subprocess(m, n_a + n_b, 4000, config4=True, first_shard=first + s0)
        batch, F = ctx.upload_flat(d, p, nr), ctx.upload_flat(fd, fp, m)
        ga, gb, fidx = np.ascontiguousarray(g[:, :n_a]), np.ascontiguousarray(g[:, n_a:]), np.arange(m)
        q = ctx.prepare_count_matrix(batch, ga, batch, gb, F, fidx, keep_per_shard=True)
        # (the FIRST execution of each slice's query — window index, shadows and program are built in front of it — is timed by itself:
        # rocprofv3 shows this launch at 22-35 ms in round 5's and round 6's traces, profiles/r06_kernel_trace_by_grid.csv)
        ctx.set_option("time_kernels", 1)
        q.run()
        torch.cuda.synchronize()
        first_run_us.append(round(ctx.get_option("last_kernel_ns") / 1e3, 1))
        ctx.set_option("time_kernels", 0)
        tot, ps = q.read(per_shard=True)
        if want_cpu:  # EVERY shard of the slice against the oracle
            t1 = time.perf_counter()
            OA, OF = PB.RowSet.from_flat(d, p, nr), PB.RowSet.from_flat(fd, fp, m)
            e = PB.count_matrix(OA, ga, OA, gb, OF, fidx)
            assert (ps == e).all(), f"config 4 strong (log-uniform rows): GPU and oracle disagree in shards {first + s0}.. of rank {rank}"
            OA.free()
            OF.free()
            t_chk += time.perf_counter() - t1
        local += tot
        nbytes += nb
        types += np.bincount(d["type"], minlength=4)[:4]
        qs.append(q)
        keep += [batch, F]
        del d, p
    resident_s = time.perf_counter() - t0
    parity = (f"every one of this rank's {ns} shards bit-exact against the oracle ({t_chk:.1f} s, {PB.threads()} host threads)" if want_cpu else "unchecked (--no-cpu-baseline)")
    expected = torch.from_numpy(local.view(np.int64).reshape(-1).copy()).to(dev)
    if world > 1:
        dist.all_reduce(expected)
    expected = expected.cpu().numpy().view(np.uint64)

    def run_local(cell):
        if not qs:
            cell.zero_()  # (fewer shards than ranks: this rank contributes nothing)
        for i, q in enumerate(qs):
            q.run(cell.data_ptr(), accumulate=i > 0)

    # the kernels alone: the sum over the slices' launches (HIP events by the library around each)
    k_us = []
    if qs:
        ctx.set_option("time_kernels", 1)
        for _ in range(5):
            t = 0.0
            for q in qs:
                q.run()
                torch.cuda.synchronize()
                t += ctx.get_option("last_kernel_ns") / 1e3
            k_us.append(t)
        ctx.set_option("time_kernels", 0)
    k_med = sorted(k_us)[len(k_us) // 2] if k_us else 0.0
    res = fdist.strong_scaling_queries(run_local, n_a * n_b, args.queries4, dev, expected=expected, sync=torch.cuda.synchronize, depth=4, cpu_group=cpu_group)
    tv = torch.tensor([res["pipelined_s_per_query"], k_med], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
    tv = tv.tolist()
    for q in qs:
        q.free()
    for b in keep:
        b.free()
    return {"id": "loguniform", "rows": f"{n_a} x {n_b} + filter row per shard, densities log-uniform [0.001, 0.5] (SURVEY 8d), encoded: {int(types[1])} array / {int(types[2])} bitmap / {int(types[3])} run containers on this rank",
            "shards_total": total, "shards_this_rank": ns, "slices": len(qs), "resident_bytes_this_rank": int(nbytes), "make_resident_s": resident_s,
            "ms_per_query_pipelined": tv[0] * 1e3, "set_ops_per_s": total * 16 * n_a * n_b / tv[0] if tv[0] else None, "kernel": "k_count_matrix_fusedq", "kernel_us": k_med,
            "kernel_us_max_over_ranks": tv[1], "kernel_frac": (nbytes / (k_med * 1e-6) / 1e9 / HBM_PEAK_GBPS) if k_med else None, "parity": parity,
            "kernel_us_first_run_of_each_slice": first_run_us,
            "collectives": res["collectives"], "latency_s": res["latency_s"]}


def group_in_process(ctx_lib_path, wa, wb, expected, steps):
    """N = 1 box: the in-library multi-device path with 2 members sharing device 0 (correctness of
    the G > 1 code path + its per-query latency; not a scaling number)."""
    from featurebase_amd import lib as L
    from featurebase_amd.roaring import Group

    grp = Group([0, 0])
    n = wa.shape[0]
    plans, keep = [], []
    for m, c in enumerate(grp.members):
        A, B = c.upload_dense(np.ascontiguousarray(wa[m::2])), c.upload_dense(np.ascontiguousarray(wb[m::2]))
        keep += [A, B]
        plans.append(c.plan(A, np.arange(len(range(m, n, 2))), B, np.arange(len(range(m, n, 2)))))
    res = {"members": 2, "devices": [0, 0], "note": "both members share device 0: exercises the G > 1 path, not a scaling measurement", "modes": {}}
    for name, mode in (("host", L.REDUCE_HOST), ("peer", L.REDUCE_PEER)):
        grp.set_reduce(mode)
        for _ in range(5):
            tot = grp.plan_intersection_count_total(plans)
        assert tot == expected, (name, tot, expected)
        lat = []
        for _ in range(steps):
            t0 = time.perf_counter()
            tot = grp.plan_intersection_count_total(plans)
            lat.append((time.perf_counter() - t0) * 1e3)
        assert tot == expected
        res["modes"][name] = {"latency_ms": dist_of(lat), "total_matches_numpy": True}
    for p in plans:
        p.free()
    for b in keep:
        b.free()
    grp.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shards", type=int, default=SHARDS_PER_GPU, help="shards per GPU")
    ap.add_argument("--repeats", type=int, default=50, help="additional timed regions of --steps steps (distribution of the step time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 3, 4, 5 (N = 1 only anyway)")
    ap.add_argument("--shards3", type=int, default=256)
    ap.add_argument("--shards4", type=int, default=1024)
    ap.add_argument("--shards4-mixed", type=int, default=1024, help="BASELINE configs[3] per-GPU slice on SURVEY 8d's log-uniform density rows (0 = skip)")
    ap.add_argument("--shards4-mixed-total", type=int, default=8192, help="strong-scaling variant of configs[3] on the log-uniform rows: this many shards in total (0 = skip)")
    ap.add_argument("--shards5", type=int, default=96)
    ap.add_argument("--secondary-iters", type=int, default=20)
    ap.add_argument("--shards4-total", type=int, default=8192, help="BASELINE configs[3], strong scaling: this many shards in total, split over the ranks (0 = skip)")
    ap.add_argument("--queries4", type=int, default=10, help="timed queries per reduce mode of the strong-scaling section")
    ap.add_argument("--detail", default="bench_detail.json", help="file name (under the repo root, and gpurun_out/ when present) of the verbose result object; stdout carries the compact line")
    ap.add_argument("--two-contexts", action="store_true",
                    help="also time two independent queries in flight (root context + fbk_ctx_fork); off by default: its overlapping launches would "
                         "double the per-launch durations a kernel trace of this command shows for the headline kernel")
    ap.add_argument("--cold-sets", type=int, default=4, help="distinct resident data sets cycled for the L3-cold roofline (1 = skip)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE line, the JSON: libraries below print to file descriptor 1 on their own (RCCL its
    # version banner at communicator creation, gloo its connection notes).  Everything but that line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line whose n_gpus differs from what was asked for")
    n_gpus = world

    # config 3's rows are generated first, in forked worker processes, BEFORE this process touches the
    # HIP runtime (a fork afterwards would be unsafe)
    pre3 = None
    if n_gpus == 1 and not args.no_secondary and args.shards3:
        import datagen as D0

        t0 = time.perf_counter()
        r3, g3, f3 = D0.config3_flat(args.shards3, mp="fork")
        pre3 = (r3, g3, f3, time.perf_counter() - t0)

    import torch  # (first: one HIP runtime per process, see featurebase_amd/lib.py)

    # one process per GPU.  (FBK_BENCH_BACKEND=gloo: fewer devices than ranks, see self_launch.)
    backend = os.environ.get("FBK_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % n_dev if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    from featurebase_amd import dist as fdist

    if n_gpus > 1:
        import torch.distributed as dist

        fdist.init(backend, dev)  # backend "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == n_gpus
    import datagen as D
    from featurebase_amd import lib as L
    from featurebase_amd.roaring import Context

    ctx = Context(dev_index)
    stream = torch.cuda.Stream(device=dev)  # non-default: its handle is what fbk launches on
    ctx.set_stream(stream.cuda_stream)

    n = args.shards
    # synthetic shards of this rank: global shard id = rank*n + s seeds the generator
    wa = D.dense_rows(n, 0.5, 1000 + 2 * rank)
    wb = D.dense_rows(n, 0.5, 1001 + 2 * rank)
    ctx.upload_dense(wa[:1]).free()  # (the context's two pinned upload buffers are allocated by its first bulk upload: not part of the rate below)
    t_up0 = time.perf_counter()
    A, B = ctx.upload_dense(wa), ctx.upload_dense(wb)
    t_upload = time.perf_counter() - t_up0
    rows = np.arange(n)

    with torch.cuda.stream(stream):
        counts = torch.zeros(n, dtype=torch.int64, device=dev)  # uint64 payload; int64 for RCCL
        total = torch.zeros(1, dtype=torch.int64, device=dev)
    plan = ctx.plan(A, rows, B, rows, device_counts_ptr=counts.data_ptr())

    # The per-node totals of consecutive steps land in consecutive slots of a small device vector
    # (featurebase_amd/dist.py BucketedCountReducer): the vector is cleared once per REDUCE_BUCKET
    # steps and, for N > 1, all-reduced over RCCL/xGMI once per REDUCE_BUCKET steps, asynchronously.
    red = fdist.BucketedCountReducer(REDUCE_BUCKET, dev)

    def step_bucketed():
        # per-shard |a ∩ b| and the per-node reduce in ONE launch: every workgroup of k_icount_dense
        # adds its count to the step's (zeroed) slot.  Measured alternatives: a second launch
        # (k_sum_u64) +2.3 us, "last workgroup sums the per-shard counts" +3.1 us per step.
        plan.intersection_count_accumulate(red.slot_ptr())
        red.advance()  # N > 1: RCCL sum of the partial counts over xGMI once the bucket is full

    # N > 1, the headline: EVERY step's partial total is all-reduced on its own (one collective per query, asynchronous on
    # the communicator's stream, PER_QUERY_RING cells rotating so that the next steps' kernels never touch a cell a
    # collective still reads).  The ring is cleared once per revolution.
    pq = fdist.PerQueryReducer(1, PER_QUERY_RING, dev)
    # The collective is issued by the LIBRARY when it can be (fbk_comm_*: its own RCCL communicator and stream; torch only carries
    # the 128-byte unique id at start-up): one torch all-reduce costs the launching thread ~28 us, more than half of a 41 us step
    # (profiles/r06_collective_host_cost.json, measured on a one-rank group).  Any rank failing to set it up — no librccl, gloo
    # ranks sharing a device — leaves every rank on torch's collectives (the cross-check path either way).
    lib_comm = False
    if n_gpus > 1 and backend == "nccl" and os.environ.get("FBK_BENCH_LIBCOMM", "1") != "0":
        lib_comm = fdist.library_comm_init(ctx)
    lpq = fdist.LibraryPerQueryReducer(ctx, 1, PER_QUERY_RING, dev) if lib_comm else None

    def step_per_query_torch():
        if pq.k % PER_QUERY_RING == 0:
            pq.flush()
            pq.buf.zero_()
        plan.intersection_count_accumulate(pq.cell().data_ptr())
        pq.reduce()

    def step_per_query_library():
        if lpq.k % PER_QUERY_RING == 0:
            lpq.flush()  # (one event: the context's stream waits for the collectives of the last revolution)
            lpq.buf.zero_()
        plan.intersection_count_accumulate(lpq.cell_ptr())
        lpq.reduce()

    step_per_query = step_per_query_library if lib_comm else step_per_query_torch

    step = step_per_query if n_gpus > 1 else step_bucketed
    flush = ((lambda: lpq.flush()) if lib_comm else (lambda: pq.flush())) if n_gpus > 1 else (lambda: red.flush())
    host_enqueue = []  # seconds the launching thread spent in each timed region's loop (before anything is waited for)

    def barrier():
        if n_gpus > 1:
            dist.barrier()

    def timed_region(k):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        host_enqueue.append((time.perf_counter() - t0) / k)
        reduced = flush()  # every outstanding collective (N = 1: nothing to wait for): inside the timed region
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, reduced

    extra = {}
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        flush()
        dt, reduced = timed_region(args.steps)  # THE timed region: exactly --steps steps

        # parity spot check of the timed result (numpy popcount of this rank's shards)
        local_expected = int(np.bitwise_count(wa & wb).sum())
        got_counts = counts.cpu().numpy().view(np.uint64)
        assert int(got_counts.sum()) == local_expected, "GPU result differs from numpy popcount"
        # every used slot must hold the sum over ranks of the per-rank expected counts
        ge = torch.tensor([local_expected], dtype=torch.int64, device=dev)
        if n_gpus > 1:
            dist.all_reduce(ge)
        global_expected = int(ge.item())
        vals = torch.cat([b.reshape(-1) for b in reduced]).cpu().numpy()
        vals = vals[vals != 0]
        assert vals.size > 0 and (vals == global_expected).all(), (
            f"reduced totals differ from the sum of the per-shard counts: expected {global_expected} (this rank {local_expected}), cells "
            f"{[int(x) for x in torch.cat([b.reshape(-1) for b in reduced]).cpu().numpy()]}")

        # ---- distribution: the same timed region `repeats` more times (median / p10 / p90 of the step time)
        rep = []
        for _ in range(args.repeats):
            d, _r = timed_region(args.steps)
            if n_gpus > 1:
                t = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                d = float(t.item())
            rep.append(d / args.steps * 1e3)

        # ---- N > 1: what ONE query pays.  (a) a collective per step, pipelined on the device;
        # (b) the same with the total read back by the host after every step (per-query latency);
        # (c) "copy the partials to the host and add": D2H of each rank's partial + a gloo all-reduce
        if n_gpus > 1:
            # the bucketed throughput mode of the earlier rounds (REDUCE_BUCKET steps per collective): NOT what a query sees
            saved = (step, flush)
            step, flush = step_bucketed, (lambda: red.flush())
            for _ in range(args.warmup):
                step()
            flush()
            bdt, _b = timed_region(args.steps)
            step, flush = saved
            tb = torch.tensor([bdt], dtype=torch.float64, device=dev)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            extra["throughput_mode_bucketed"] = {"ms_per_step": float(tb.item()) / args.steps * 1e3, "set_ops_per_s": n_gpus * n * 16 * args.steps / float(tb.item()),
                                                 "steps_per_collective": REDUCE_BUCKET, "note": "one all-reduce per 16 steps, asynchronous: a throughput mode no single query sees; the headline reduces every step on its own"}
            lat_steps = min(args.steps, 200)

            def step_collective():
                plan.intersection_count_total(total.data_ptr())
                dist.all_reduce(total)

            for _ in range(5):
                step_collective()
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(lat_steps):
                step_collective()
            torch.cuda.synchronize()
            barrier()
            pipelined = (time.perf_counter() - t0) / lat_steps * 1e3
            assert int(total.item()) == global_expected
            lat = []
            for _ in range(lat_steps):
                t0 = time.perf_counter()
                step_collective()
                v = int(total.item())
                lat.append((time.perf_counter() - t0) * 1e3)
            assert v == global_expected
            host_lat = None
            try:
                cpu_group = dist.new_group(backend="gloo")
                pinned = torch.zeros(1, dtype=torch.int64).pin_memory()
                hl = []
                for _ in range(lat_steps):
                    t0 = time.perf_counter()
                    plan.intersection_count_total(total.data_ptr())
                    pinned.copy_(total, non_blocking=True)
                    torch.cuda.synchronize()
                    dist.all_reduce(pinned, group=cpu_group)
                    hl.append((time.perf_counter() - t0) * 1e3)
                assert int(pinned.item()) == global_expected
                host_lat = dist_of(hl)
            except Exception as e:  # noqa: BLE001
                host_lat = {"error": str(e)}
            extra["per_query"] = {
                "collective_per_step_pipelined_ms_per_step": pipelined,
                "collective_per_step_set_ops_per_s": n_gpus * n * 16 / (pipelined * 1e-3),
                "collective_per_step_host_readback_latency_ms": dist_of(lat),
                "host_add_latency_ms": host_lat,
                "note": "the headline already carries one all-reduce per step (pipelined over a ring of cells); here the same with ONE cell (each step waits for the previous collective) and with the total read back by the host after every step",
            }

        # ---- roofline of the dominant kernel: HIP events around back-to-back launches
        # of k_icount_dense alone, on the stream it is launched on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kiters = max(args.steps, 50)
        kernel_step = (lambda: plan.intersection_count_accumulate(total.data_ptr()))  # the launch the timed steps make
        for _ in range(5):
            kernel_step()
        torch.cuda.synchronize()
        k_samples = []
        for _ in range(max(10, min(args.repeats, 50))):
            e0.record(stream)
            for _ in range(kiters):
                kernel_step()
            e1.record(stream)
            torch.cuda.synchronize()
            k_samples.append(e0.elapsed_time(e1) / kiters)
        k_dist = dist_of(k_samples)
        k_ms = k_dist["median"]
        alg_bytes = 2 * n * 16 * 8192 + n * 8  # both operands read once + one u64 count per shard
        # ---- the same kernel with the Infinity Cache taken out of the picture: the 256 MiB
        # working set of configs[1] is exactly the size of the 256 MiB L3, so cycle over
        # several distinct resident data sets (cold_sets x 256 MiB) between launches
        cold = None
        if args.cold_sets > 1:
            xs = []
            for i in range(1, args.cold_sets):
                xa = ctx.upload_dense(D.dense_rows(n, 0.5, 5000 + 2 * i + 100 * rank))
                xb = ctx.upload_dense(D.dense_rows(n, 0.5, 5001 + 2 * i + 100 * rank))
                xs.append((xa, xb, ctx.plan(xa, rows, xb, rows)))
            plans = [plan] + [e[2] for e in xs]
            for i in range(2 * len(plans)):
                plans[i % len(plans)].intersection_count()
            torch.cuda.synchronize()
            citers = (kiters // len(plans)) * len(plans)
            e0.record(stream)
            for i in range(citers):
                plans[i % len(plans)].intersection_count()
            e1.record(stream)
            torch.cuda.synchronize()
            c_ms = e0.elapsed_time(e1) / citers
            cold = {
                "kernel": "k_icount_dense<16>",
                "working_set_MiB": len(plans) * 2 * n * 16 * 8192 / 2**20,
                "achieved": alg_bytes / (c_ms * 1e-3) / 1e9,
                "unit": "GB/s",
                "frac": alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel_us": c_ms * 1e3,
            }
            for xa, xb, pl in xs:
                pl.free()
                xa.free()
                xb.free()
        # ---- two independent queries at a time: a forked context (fbk_ctx_fork: its own stream, lock and pool — the analogue of
        # the reference's pool of shard workers, executor.go:6723-6737) runs the same step beside the root context, the launches
        # alternating.  NOT the headline (a step there is one launch behind the other on one stream): it shows what the dispatch gap
        # between consecutive launches of ONE stream is worth — the drain of one query's launch overlaps the ramp of the other's.
        # Measured (round 6, profiles/r06_two_contexts.json): 40.55 against 41.06 us per step, 39.81 against 41.15 under rocprofv3 —
        # 1-3 %: the gap is not idle memory time a second stream can fill.  Opt-in (--two-contexts).
        two_ctx = None
        try:
            if n_gpus > 1 or not args.two_contexts:
                raise RuntimeError("N = 1 with --two-contexts only")
            ctx2 = ctx.fork()
            stream2 = torch.cuda.Stream(device=dev)
            ctx2.set_stream(stream2.cuda_stream)
            counts2 = torch.zeros(n, dtype=torch.int64, device=dev)
            cells = torch.zeros(2, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            plan2 = ctx2.plan(A, rows, B, rows, device_counts_ptr=counts2.data_ptr())
            c0, c1 = cells.data_ptr(), cells.data_ptr() + 8
            for _ in range(10):
                plan.intersection_count_accumulate(c0)
                plan2.intersection_count_accumulate(c1)
            torch.cuda.synchronize()
            cells.zero_()
            torch.cuda.synchronize()
            t_regions = []
            for _ in range(9):
                t0 = time.perf_counter()
                for _ in range(kiters):
                    plan.intersection_count_accumulate(c0)
                    plan2.intersection_count_accumulate(c1)
                torch.cuda.synchronize()
                t_regions.append((time.perf_counter() - t0) / (2 * kiters))
            got2 = cells.cpu().numpy()
            assert int(got2[0]) == 9 * kiters * local_expected and int(got2[1]) == 9 * kiters * local_expected, "two-context totals differ"
            t_regions.sort()
            t2 = t_regions[len(t_regions) // 2]
            two_ctx = {"ms_per_step": t2 * 1e3, "set_ops_per_s": n * 16 / t2, "frac_of_8TBps": alg_bytes / t2 / 1e9 / HBM_PEAK_GBPS, "contexts": 2,
                       "note": "two independent queries in flight (root context + fbk_ctx_fork, one stream each), launches alternating; wall clock over "
                               f"{2 * kiters} steps, median of 9 regions, totals checked; the headline runs its steps one behind the other on ONE stream"}
            plan2.free()
            ctx2.close()
        except Exception as e:  # noqa: BLE001 — a secondary figure never fails the run
            two_ctx = {"error": str(e)[:200]} if (n_gpus == 1 and args.two_contexts) else None
        # ---- materialising variant: Intersect written out + Count fused (roaring.go:4960)
        for _ in range(5):
            plan.setop(L.OP_AND)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(kiters):
            plan.setop(L.OP_AND)
        e1.record(stream)
        torch.cuda.synchronize()
        m_ms = e0.elapsed_time(e1) / kiters
        m_bytes = 3 * n * 16 * 8192 + n * 16 * 16 + n * 8
        plan.total(total.data_ptr())
        torch.cuda.synchronize()
        assert int(total.item()) == local_expected

        strong = None
        if args.shards4_total:
            try:
                cpu_group = dist.new_group(backend="gloo") if n_gpus > 1 else None
                mine4 = config4_strong(torch, dist if n_gpus > 1 else None, fdist, dev, ctx, stream, rank, n_gpus, args, cpu_group, not args.no_cpu_baseline)
                # max over ranks of every time; the reduced matrix was checked against the sum of the ranks' own results in every mode
                keys = [("pipelined_s_per_query", None), ("latency_s", "median"), ("host_add_latency_s", "median")]
                tv = torch.tensor([mine4[k] if sub is None else mine4[k][sub] for k, sub in keys] + [mine4["kernel_us"]["median"]], dtype=torch.float64, device=dev)
                if n_gpus > 1:
                    dist.all_reduce(tv, op=dist.ReduceOp.MAX)
                tv = tv.tolist()
                total_ops = args.shards4_total * 16 * 32 * 32
                strong = {
                    "workload": f"configs[3]: {args.shards4_total} shards x (32 x 32 rows + filter), dense, IntersectionCount matrix; shard s on rank s mod N; one all-reduce of the 1024-cell partial matrix per query",
                    "scaling": "strong",
                    "n_gpus": n_gpus,
                    "backend": (("rccl" if backend == "nccl" else backend) if n_gpus > 1 else None),
                    "ms_per_query_pipelined": tv[0] * 1e3,
                    "set_ops_per_s": total_ops / tv[0],
                    "ms_per_query_latency_host_readback": tv[1] * 1e3,
                    "ms_per_query_host_add": tv[2] * 1e3,
                    "kernel_us_max_over_ranks": tv[3],
                    "rank0": mine4,
                }
                vs = mine4.pop("variants", [])
                vs[0].update(ms_per_query_pipelined=tv[0] * 1e3, set_ops_per_s=total_ops / tv[0], kernel_us_max_over_ranks=tv[3])
                strong["variants"] = vs
                if len(vs) > 1:
                    strong["workload"] += "; variant 'loguniform': the same query on SURVEY 8d's log-uniform density rows (encoded array / bitmap rows)"
            except Exception as e:  # noqa: BLE001 — the headline is measured already
                import traceback

                strong = {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc().splitlines()[-6:]}
                if n_gpus > 1:
                    raise  # a rank that left the section early would leave the others in a collective

        secondary = None
        if n_gpus == 1 and not args.no_secondary:
            t_s0 = time.perf_counter()
            try:
                secondary = secondary_configs(torch, dev, ctx, stream, pre3, args, not args.no_cpu_baseline)
            except Exception as e:  # the headline above is measured already: a failing secondary entry must not cost the bench line
                import traceback

                secondary = [{"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc().splitlines()[-6:]}]
            extra["secondary_wall_s"] = time.perf_counter() - t_s0

    # max over ranks
    if n_gpus > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the in-library multi-device path (one process, fbk_group_*): rank 0 runs it in a child process
    # over the same devices while the other ranks wait at the barrier below
    group_api = None
    if n_gpus > 1 and rank == 0:
        devs = ",".join(str(r if backend == "nccl" else r % n_dev) for r in range(n_gpus))
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK")}
            p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "group_bench.py"), "--devices", devs, "--shards", str(n), "--steps", str(min(args.steps, 200)),
                                "--matrix", "32", "--matrix-total-shards", str(args.shards4_total)],
                               capture_output=True, text=True, timeout=300, env=env)
            line = [x for x in p.stdout.splitlines() if x.startswith("{")]
            group_api = json.loads(line[-1]) if line else {"error": (p.stderr or "no output")[-400:]}
        except subprocess.TimeoutExpired as e:  # a reduce mode hung (the script prints a line per finished mode: keep those)
            so = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            line = [x for x in so.splitlines() if x.startswith("{")]
            group_api = json.loads(line[-1]) if line else {}
            group_api["error"] = "scripts/group_bench.py did not finish within 300 s (killed); modes listed are the ones that completed"
        except Exception as e:  # noqa: BLE001
            group_api = {"error": str(e)}
    elif n_gpus == 1:
        try:
            group_api = group_in_process(None, wa, wb, local_expected, 50)
        except Exception as e:  # noqa: BLE001
            group_api = {"error": str(e)}
    # ---- the CPU path beside it, at EVERY N (north_star: "alongside the reference Go CPU path timed on the same box's
    # host cores"): rank 0 times it on its own 1024 shards — the N = 1 workload — while the other ranks SLEEP in a gloo
    # barrier (sockets; an RCCL barrier would have every waiting rank spin on a host core the baseline is using)
    cb = None
    if not args.no_cpu_baseline:
        wait_group = None
        if n_gpus > 1:
            try:
                wait_group = dist.new_group(backend="gloo")
            except Exception:  # noqa: BLE001 — no gloo: the ranks wait in the RCCL barrier below instead
                wait_group = None
        if rank == 0:
            cb = cpu_baseline(wa, wb)
            assert cb.pop("total_count") == local_expected, "oracle and GPU disagree"
            if n_gpus > 1:
                cb["sample"] += f"; timed on rank 0 (its {n} shards = the N = 1 workload) after the timed GPU regions, the other {n_gpus - 1} ranks asleep in a host barrier"
        if wait_group is not None:
            dist.barrier(group=wait_group)
    if n_gpus > 1:
        dist.barrier()

    if rank == 0:
        # the headline: the MEDIAN of the timed regions of exactly --steps steps each (the first one, whose result is parity-checked
        # above, and the --repeats further ones; every region bracketed by barrier + synchronize, max over ranks): with the driver's
        # --steps 20 one region is 0.8 ms of wall clock, and a single sample of that moved by 1-2 % from run to run
        regions_ms = sorted([dt / args.steps * 1e3] + list(rep))
        ms_per_step = regions_ms[len(regions_ms) // 2] if len(regions_ms) % 2 else 0.5 * (regions_ms[len(regions_ms) // 2 - 1] + regions_ms[len(regions_ms) // 2])
        henq = sorted(host_enqueue)
        out = {
            "metric": "container set-ops/sec + bits-scanned GB/s, 1M-col Intersect+Count",
            "value": n_gpus * n * 16 / (ms_per_step * 1e-3),
            "unit": "set-ops/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_first_region": dt / args.steps * 1e3,
            "timed_regions": len(regions_ms),
            "host_enqueue_us_per_step": henq[len(henq) // 2] * 1e6 if henq else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: per GPU 1024 shards x 2 rows x 2^20 cols, bitmap x bitmap AND+popcount, density 50%",
                "shards_per_gpu": n,
                "containers_per_gpu": 2 * n * 16,
                "op": "Count(Intersect(Row,Row)) as IntersectionCount + per-node sum"
                + (f" + one {'RCCL' if backend == 'nccl' else backend} all-reduce of the partial total PER STEP (asynchronous, {PER_QUERY_RING} result cells in rotation)" if n_gpus > 1 else ""),
                "collectives_per_step": 1 if n_gpus > 1 else 0,
                "collective_path": (("library-rccl (fbk_comm_all_reduce_u64)" if lib_comm else "torch.distributed") if n_gpus > 1 else None),
                "parallelism": f"shards/{n_gpus}gpu",
                "backend": (("rccl" if backend == "nccl" else backend) if n_gpus > 1 else None),
                "ranks": n_gpus,
                "oversubscribed": bool(n_gpus > 1 and backend != "nccl"),
            },
            "ms_per_step_distribution": dict(dist_of(rep), note=f"{args.repeats} further timed regions of {args.steps} steps each") if rep else None,
            "bits_scanned_GBps": n_gpus * 2 * n * 16 * 8192 / (ms_per_step * 1e-3) / 1e9,
            "roofline": {
                "kernel": "k_icount_dense<16>",
                "bound": "hbm",
                "achieved": alg_bytes / (k_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "traffic": None,
                "traffic_source": None,
                "kernel_us": k_ms * 1e3,
                "kernel_us_distribution": {k: (v * 1e3 if k != "n" else v) for k, v in k_dist.items()},
                "algorithmic_bytes": alg_bytes,
            },
            "materialized": {
                "kernel": "k_setop_dense<AND>",
                "set_ops_per_s": n * 16 / (m_ms * 1e-3),
                "achieved_GBps": m_bytes / (m_ms * 1e-3) / 1e9,
                "frac": m_bytes / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel_us": m_ms * 1e3,
                "algorithmic_bytes": m_bytes,
            },
            "roofline_l3_cold": cold,
            "two_contexts": two_ctx,
            "h2d_upload_s": t_upload,
            "h2d_upload_GBps": 2 * n * 16 * 8192 / t_upload / 1e9,
            "h2d_upload_note": "fbk_batch_upload_dense of both operands from pageable numpy memory, end to end (two pinned buffers filled by host threads while the other's DMA runs, recount kernel, descriptor read-back, synchronisation); the pinned buffers exist already (a one-row upload before the clock starts)",
            "group_api": group_api,
        }
        out.update(extra)
        if secondary is not None:
            out["secondary"] = secondary
        if strong is not None:
            out["strong_scaling"] = strong
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                out["roofline"]["traffic"] = json.load(open(traffic_file)).get("k_icount_dense_hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes of an earlier run of this command, not measured in this run)"
            except Exception:
                pass
        if cb is not None:
            out["cpu_baseline"] = cb
        # stdout: ONE compact line (bench_line.py: <= 6 KB, the driver could not parse round 4's 21.6 KB line); the full
        # object goes to bench_detail.json (beside this file and under gpurun_out/) and to stderr
        import bench_line

        out["detail_files"] = bench_line.write_detail(out, ROOT, args.detail)
        print("[bench] detail: " + json.dumps(out), file=sys.stderr, flush=True)
        sys.stdout.flush()
        os.write(json_fd, (bench_line.dumps_line(out, args.detail) + "\n").encode())

    plan.free()
    A.free()
    B.free()
    ctx.close()
    if n_gpus > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
