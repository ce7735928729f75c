#!/usr/bin/env python3
"""What ONE collective per headline step costs the launching thread — measured on ONE GPU, before an 8-GPU node does it for us.

The N > 1 headline of bench.py issues one all-reduce per 41 us step (PerQueryReducer: a cell per query, asynchronous on RCCL's
stream).  Whether that loop is host-bound does not depend on the number of ranks: the launching thread pays the same torch / RCCL
call path for a one-rank communicator.  This script runs configs[1]'s step (1024 shards, bitmap x bitmap AND + popcount) in
four loops on one device and reports ms per step of each, wall clock over `--steps` steps (kernels and collectives drained):

  kernel_only            plan.intersection_count_accumulate(cell)                       (what N = 1 times)
  torch_nccl_per_step    the same + dist.all_reduce(cell, async_op=True) on a ONE-RANK nccl process group, ring of cells
                         (exactly the N > 1 headline's loop)
  host_enqueue_only      the torch loop's host time per step: perf_counter around the loop WITHOUT the final synchronize
  library_comm_per_step  the headline's loop with the all-reduce issued by the library: fbk_comm_all_reduce_u64 on the context's
                         own RCCL communicator and stream (no torch call on the hot loop)
  library_rccl_per_step  fbk_group_plan_intersection_count_total with FBK_REDUCE_RCCL on a group of one member: the kernel, then
                         ncclAllReduce through the library's dlopen'ed RCCL on the context's own stream, total read back (a
                         synchronous per-query call: includes the D2H of the total)
  library_host_per_step  the same with FBK_REDUCE_HOST (no collective: D2H + host add)

    python scripts/collective_host_cost.py [--shards 1024] [--steps 2000] [--out profiles/r06_collective_host_cost.json]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--ring", type=int, default=64)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist

    import datagen as D
    from featurebase_amd import dist as fdist
    from featurebase_amd import lib as L
    from featurebase_amd.roaring import Context, Group

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    n = args.shards
    wa, wb = D.dense_rows(n, 0.5, 1000), D.dense_rows(n, 0.5, 1001)
    expected = int(np.bitwise_count(wa & wb).sum())
    stream = torch.cuda.Stream(device=dev)
    res = {"shards": n, "steps": args.steps, "ring": args.ring, "expected_total": expected}
    with torch.cuda.stream(stream):
        ctx = Context(0)
        ctx.set_stream(stream.cuda_stream)
        A, B = ctx.upload_dense(wa), ctx.upload_dense(wb)
        rows = np.arange(n)
        counts = torch.zeros(n, dtype=torch.int64, device=dev)
        plan = ctx.plan(A, rows, B, rows, device_counts_ptr=counts.data_ptr())

        def loop(step, flush, k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step()
            t_host = time.perf_counter() - t0
            flush()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k * 1e3, t_host / k * 1e3

        # kernel only (one cell, never reduced)
        cell = torch.zeros(1, dtype=torch.int64, device=dev)
        k_only = lambda: plan.intersection_count_accumulate(cell.data_ptr())  # noqa: E731
        loop(k_only, lambda: None, 50)
        ms, host = loop(k_only, lambda: None, args.steps)
        res["kernel_only"] = {"ms_per_step": ms, "host_enqueue_ms_per_step": host}

        # the N > 1 headline's loop on a one-rank nccl group
        pq = fdist.PerQueryReducer(1, args.ring, dev, always=True)

        def step_pq():
            if pq.k % args.ring == 0:
                pq.flush()
                pq.buf.zero_()
            plan.intersection_count_accumulate(pq.cell().data_ptr())
            pq.reduce()

        loop(step_pq, pq.flush, 2 * args.ring)
        ms, host = loop(step_pq, pq.flush, args.steps)
        vals = pq.flush().reshape(-1).cpu().numpy()
        assert (vals[vals != 0] == expected).all(), "reduced totals differ"
        res["torch_nccl_per_step"] = {"ms_per_step": ms, "host_enqueue_ms_per_step": host, "collectives": pq.collectives}

        # the same loop with the collective issued by the LIBRARY (fbk_comm_all_reduce_u64: its own communicator and stream)
        if fdist.library_comm_init(ctx):
            lpq = fdist.LibraryPerQueryReducer(ctx, 1, args.ring, dev)

            def step_lib():
                if lpq.k % args.ring == 0:
                    lpq.flush()
                    lpq.buf.zero_()
                plan.intersection_count_accumulate(lpq.cell_ptr())
                lpq.reduce()

            loop(step_lib, lpq.flush, 2 * args.ring)
            ms, host = loop(step_lib, lpq.flush, args.steps)
            lpq.flush()
            torch.cuda.synchronize()
            vals = lpq.buf.reshape(-1).cpu().numpy()
            assert (vals[vals != 0] == expected).all(), "reduced totals differ (library communicator)"
            res["library_comm_per_step"] = {"ms_per_step": ms, "host_enqueue_ms_per_step": host, "collectives": lpq.collectives}
            ctx.comm_close()
        else:
            res["library_comm_per_step"] = {"error": "fbk_comm_init failed"}

        # a bare all_reduce per step (no kernel): the call path alone
        t = torch.zeros(1, dtype=torch.int64, device=dev)
        works = []

        def step_ar():
            works.append(dist.all_reduce(t, async_op=True))
            if len(works) >= args.ring:
                for w in works:
                    w.wait()
                works.clear()

        def flush_ar():
            for w in works:
                w.wait()
            works.clear()

        loop(step_ar, flush_ar, 2 * args.ring)
        ms, host = loop(step_ar, flush_ar, args.steps)
        res["torch_nccl_all_reduce_alone"] = {"ms_per_step": ms, "host_enqueue_ms_per_step": host}
        plan.free()
        A.free()
        B.free()
        ctx.close()

    # the library's own reduce on a group of one member (synchronous per call: kernel -> ncclAllReduce on the member's stream -> D2H)
    grp = Group([0])
    c = grp.members[0]
    A, B = c.upload_dense(wa), c.upload_dense(wb)
    plan = c.plan(A, np.arange(n), B, np.arange(n))
    for name, mode in (("library_host_per_step", L.REDUCE_HOST), ("library_rccl_per_step", L.REDUCE_RCCL)):
        grp.set_reduce(mode)
        for _ in range(20):
            assert grp.plan_intersection_count_total([plan]) == expected
        k = max(200, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(k):
            grp.plan_intersection_count_total([plan])
        res[name] = {"ms_per_call": (time.perf_counter() - t0) / k * 1e3, "calls": k, "note": "synchronous: includes the D2H of the total"}
    plan.free()
    A.free()
    B.free()
    grp.close()
    k0 = res["kernel_only"]["ms_per_step"]
    res["summary"] = {
        "kernel_ms": k0,
        "per_step_with_one_rank_all_reduce_ms": res["torch_nccl_per_step"]["ms_per_step"],
        "collective_adds_ms": res["torch_nccl_per_step"]["ms_per_step"] - k0,
        "host_enqueue_ms_per_step_with_collective": res["torch_nccl_per_step"]["host_enqueue_ms_per_step"],
        "host_bound": res["torch_nccl_per_step"]["host_enqueue_ms_per_step"] > k0,
        "library_comm_per_step_ms": res["library_comm_per_step"].get("ms_per_step"),
        "library_comm_host_enqueue_ms_per_step": res["library_comm_per_step"].get("host_enqueue_ms_per_step"),
    }
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
