#!/usr/bin/env python3
"""Single-process multi-GPU measurement of the in-library group path (fbk_group_*, include/fbk.h).

One process owns G devices (a Go server's deployment): member m holds the shards s with
s % G == m, one step = Count(Intersect(Row, Row)) over every shard of every member
(fbk_group_plan_intersection_count_total: one launch per device, all devices concurrently, then the
reduce of the G partial counts).  Every reduce mode the devices allow is timed —
"host" (copy G partials to pinned host memory, add there), "peer" (one kernel on member 0 reading
the other devices over xGMI), "rccl" (ncclAllReduce over single-process communicators) — as the
per-query latency a caller sees (the total is back on the host after every step).

    python scripts/group_bench.py --devices 0,1,2,3,4,5,6,7 [--shards 1024] [--steps 200]

The same device may be listed several times (members share it): that exercises the complete G > 1
path on a one-GPU box; RCCL is skipped then.  Prints one JSON line per finished reduce mode, each a superset of
the one before (the LAST line is the result).  bench.py runs this script from rank 0 (N > 1) and embeds
the last line as "group_api".
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
    ap.add_argument("--devices", default="0")
    ap.add_argument("--shards", type=int, default=1024, help="shards per member (weak scaling)")
    ap.add_argument("--total-shards", type=int, default=0, help="fixed total, split over the members (strong scaling)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--matrix", type=int, default=0, help="also time a GroupBy count matrix of this many rows per side (dense rows)")
    ap.add_argument("--matrix-shards", type=int, default=32, help="shards per member for --matrix")
    args = ap.parse_args()
    devices = [int(x) for x in args.devices.split(",")]
    G = len(devices)

    import datagen as D
    from featurebase_amd import lib as L
    from featurebase_amd.roaring import Group

    grp = Group(devices)
    per = [args.shards] * G
    if args.total_shards:
        per = [len(range(m, args.total_shards, G)) for m in range(G)]
    batches, plans, expected = [], [], 0
    for m, ctx in enumerate(grp.members):
        n = per[m]
        if n == 0:
            plans.append(None)
            continue
        wa, wb = D.dense_rows(n, 0.5, 7000 + 2 * m), D.dense_rows(n, 0.5, 7001 + 2 * m)
        expected += int(np.bitwise_count(wa & wb).sum())
        A, B = ctx.upload_dense(wa), ctx.upload_dense(wb)
        batches += [A, B]
        plans.append(ctx.plan(A, np.arange(n), B, np.arange(n)))
    modes = [("host", L.REDUCE_HOST), ("peer", L.REDUCE_PEER)]
    if len(set(devices)) == G and G > 1:
        modes.append(("rccl", L.REDUCE_RCCL))
    out = {
        "members": G,
        "devices": devices,
        "distinct_devices": len(set(devices)),
        "shards_per_member": per,
        "op": "fbk_group_plan_intersection_count_total: Count(Intersect(Row,Row)) over all members' shards, total on the host after every step",
        "modes": {},
    }
    set_ops_per_step = sum(per) * 16
    for name, mode in modes:
        try:
            grp.set_reduce(mode)
        except L.FbkError as e:
            out["modes"][name] = {"error": str(e)}
            continue
        for _ in range(args.warmup):
            tot = grp.plan_intersection_count_total(plans)
        assert tot == expected, (name, tot, expected)
        lat = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t1 = time.perf_counter()
            tot = grp.plan_intersection_count_total(plans)
            lat.append(time.perf_counter() - t1)
        dt = time.perf_counter() - t0
        assert tot == expected, (name, tot, expected)
        lat.sort()
        out["modes"][name] = {
            "ms_per_step": dt / args.steps * 1e3,
            "latency_ms_median": lat[len(lat) // 2] * 1e3,
            "latency_ms_p10": lat[len(lat) // 10] * 1e3,
            "latency_ms_p90": lat[(len(lat) * 9) // 10] * 1e3,
            "set_ops_per_s": set_ops_per_step * args.steps / dt,
            "bits_scanned_GBps": 2 * set_ops_per_step * 8192 * args.steps / dt / 1e9,
            "total_matches_numpy": True,
        }
        print(json.dumps(out), flush=True)  # (a line per finished mode: a caller that has to kill a hung mode keeps the earlier ones)
    for p in plans:
        if p is not None:
            p.free()
    for b in batches:
        b.free()
    if args.matrix:
        n_a = n_b = args.matrix
        ns = args.matrix_shards
        per_member, keep, exp = [], [], np.zeros((n_a, n_b), dtype=np.uint64)
        for m, ctx in enumerate(grp.members):
            wa, wb, wf = D.dense_rows(ns * n_a, 0.5, 7100 + 3 * m), D.dense_rows(ns * n_b, 0.5, 7101 + 3 * m), D.dense_rows(ns, 0.5, 7102 + 3 * m)
            A, B, F = ctx.upload_dense(wa), ctx.upload_dense(wb), ctx.upload_dense(wf)
            keep += [A, B, F]
            per_member.append(dict(a=A, rows_a=np.arange(ns * n_a).reshape(ns, n_a), b=B, rows_b=np.arange(ns * n_b).reshape(ns, n_b), filt=F, rows_f=np.arange(ns)))
            for s in range(ns):  # spot check of one cell per member
                exp[1, 2] += np.bitwise_count(wa[s * n_a + 1] & wb[s * n_b + 2] & wf[s]).sum()
        mm = {}
        for name, mode in modes:
            try:
                grp.set_reduce(mode)
            except L.FbkError as e:
                mm[name] = {"error": str(e)}
                continue
            for _ in range(3):
                tot = grp.count_matrix(per_member, n_a, n_b)
            assert int(tot[1, 2]) == int(exp[1, 2]), (name, int(tot[1, 2]), int(exp[1, 2]))
            t0 = time.perf_counter()
            it = max(5, args.steps // 10)
            for _ in range(it):
                grp.count_matrix(per_member, n_a, n_b)
            dt = (time.perf_counter() - t0) / it
            mm[name] = {"ms_per_call": dt * 1e3, "container_pairs_per_s": G * ns * 16 * n_a * n_b / dt}
        out["count_matrix"] = {"rows_per_side": args.matrix, "shards_per_member": ns, "modes": mm}
        for b in keep:
            b.free()
    grp.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
