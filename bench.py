#!/usr/bin/env python3
"""bench.py — headline benchmark of the roaring set-op hot path on MI355X.

Workload (BASELINE.json configs[1]): per GPU 1024 shards x 2 rows x 2^20 columns, density
50 % => 32768 bitmap containers (256 MiB) resident in HBM; one *step* = one pass of
Count(Intersect(Row a, Row b)) over every shard: |a_s ∩ b_s| for each shard s
(Bitmap.IntersectionCount, roaring.go:711), the per-node sum (executeCount reduceFn,
executor.go:5880) and, for N > 1, the cross-GPU sum of the partial counts by one RCCL
all-reduce (the exchange step executor.go:6449 mapReduce does over HTTP).

    python bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = container set-ops per second over all N GPUs
(a set-op = one container pair with equal keys); weak scaling (1024 shards
per GPU).  Also reported: bits-scanned GB/s, the roofline of the dominant kernel (HIP
events around back-to-back launches), the materialising variant, and the CPU oracle timed
on this box's host cores as `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: one HIP runtime per process, see featurebase_amd/lib.py)

SHARDS_PER_GPU = 1024
REDUCE_BUCKET = 16  # steps per RCCL all-reduce when N > 1
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


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
    return {
        "value": n * 16 * pm / tm,
        "unit": "set-ops/s",
        "cores": cores,
        "kind": "port",
        "sample": f"full workload ({n} shards x 2 rows, 256 MiB) x {pm} passes on {cores} threads ({tm:.1f} s), "
        f"C restatement of the Go path (oracle/roaring_oracle.c) built with {build}; each thread re-scans its own "
        f"{max(1, n // cores)}-shard chunk ({max(1, n // cores) * 256} KiB), i.e. the CPU figure is cache-resident: an upper bound for the CPU",
        "bits_scanned_GBps": 2 * n * 16 * 8192 * pm / tm / 1e9,
        "single_thread_set_ops_per_s": single,
        "single_thread_no_autovectorize_set_ops_per_s": novec,
        "total_count": int(tot),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shards", type=int, default=SHARDS_PER_GPU, help="shards per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cold-sets", type=int, default=4, help="distinct resident data sets cycled for the L3-cold roofline (1 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = max(world, 1)
    # one process per GPU.  (FBK_BENCH_BACKEND=gloo lets the N > 1 code path be smoke-tested on
    # a single-GPU box: every rank then shares device 0 and the collectives go through gloo.)
    backend = os.environ.get("FBK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    from featurebase_amd import dist as fdist

    if n_gpus > 1:
        import torch.distributed as dist

        fdist.init(backend, dev)  # backend "nccl" IS RCCL on ROCm

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

    def step():
        # per-shard |a ∩ b| and the per-node reduce in ONE launch: every workgroup of k_icount_dense
        # adds its count to the step's (zeroed) slot.  Measured alternatives: a second launch
        # (k_sum_u64) +2.3 us, "last workgroup sums the per-shard counts" +3.1 us per step.
        plan.intersection_count_accumulate(red.slot_ptr())
        red.advance()  # N > 1: RCCL sum of the partial counts over xGMI once the bucket is full

    def barrier():
        if n_gpus > 1:
            dist.barrier()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        red.flush()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        reduced = red.flush()  # the tail bucket + every outstanding collective: inside the timed region
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0

        # parity spot check of the timed result (numpy popcount of this rank's shards)
        local_expected = int(np.bitwise_count(wa & wb).sum())
        got_counts = counts.cpu().numpy().view(np.uint64)
        assert int(got_counts.sum()) == local_expected, "GPU result differs from numpy popcount"
        # every used slot must hold the sum over ranks of the per-rank expected counts
        ge = torch.tensor([local_expected], dtype=torch.int64, device=dev)
        if n_gpus > 1:
            dist.all_reduce(ge)
        vals = torch.cat([b for b in reduced]).cpu().numpy()
        vals = vals[vals != 0]
        assert vals.size > 0 and (vals == int(ge.item())).all(), "reduced totals differ from the sum of the per-shard counts"

        # ---- roofline of the dominant kernel: HIP events around back-to-back launches
        # of k_icount_dense alone, on the stream it is launched on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kiters = max(args.steps, 50)
        kernel_step = (lambda: plan.intersection_count_accumulate(total.data_ptr()))  # the launch the timed steps make
        for _ in range(5):
            kernel_step()
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(kiters):
            kernel_step()
        e1.record(stream)
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / kiters
        alg_bytes = 2 * n * 16 * 8192 + n * 8  # both operands read once + one u64 count per shard
        # ---- the same kernel with the Infinity Cache taken out of the picture: the 256 MiB
        # working set of configs[1] is exactly the size of the 256 MiB L3, so cycle over
        # several distinct resident data sets (cold_sets x 256 MiB) between launches
        cold = None
        if args.cold_sets > 1:
            extra = []
            for i in range(1, args.cold_sets):
                xa = ctx.upload_dense(D.dense_rows(n, 0.5, 5000 + 2 * i + 100 * rank))
                xb = ctx.upload_dense(D.dense_rows(n, 0.5, 5001 + 2 * i + 100 * rank))
                extra.append((xa, xb, ctx.plan(xa, rows, xb, rows)))
            plans = [plan] + [e[2] for e in extra]
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
            for xa, xb, pl in extra:
                pl.free()
                xa.free()
                xb.free()
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

    # max over ranks
    if n_gpus > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        set_ops = n_gpus * n * 16 * args.steps
        ms_per_step = dt / args.steps * 1e3
        out = {
            "metric": "container set-ops/sec + bits-scanned GB/s, 1M-col Intersect+Count",
            "value": set_ops / dt,
            "unit": "set-ops/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
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
                + (f" + RCCL all-reduce of the partial totals ({REDUCE_BUCKET} steps per collective, async)" if n_gpus > 1 else ""),
                "parallelism": f"shards/{n_gpus}gpu",
            },
            "bits_scanned_GBps": n_gpus * 2 * n * 16 * 8192 / (dt / args.steps) / 1e9,
            "roofline": {
                "kernel": "k_icount_dense<16>",
                "bound": "hbm",
                "achieved": alg_bytes / (k_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "traffic": None,
                "kernel_us": k_ms * 1e3,
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
            "h2d_upload_s": t_upload,
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                out["roofline"]["traffic"] = json.load(open(traffic_file)).get("k_icount_dense_hbm_bytes_per_launch")
            except Exception:
                pass
        if n_gpus == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(wa, wb)
            assert cb.pop("total_count") == local_expected, "oracle and GPU disagree"
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)

    plan.free()
    A.free()
    B.free()
    ctx.close()
    if n_gpus > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
