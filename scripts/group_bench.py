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
    ap.add_argument("--matrix-total-shards", type=int, default=0, help="--matrix over this many shards in TOTAL, split over the members (strong scaling: BASELINE configs[3] with 8192), rows generated on the devices")
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
        import torch

        n_a = n_b = args.matrix
        strong = args.matrix_total_shards > 0
        per_member, keep, exp = [], [], np.zeros((n_a, n_b), dtype=np.uint64)
        shards_of = [len(range(m, args.matrix_total_shards, G)) if strong else args.matrix_shards for m in range(G)]
        for m, ctx in enumerate(grp.members):
            ns = shards_of[m]
            if ns == 0:
                per_member.append(None)
                continue
            if strong:  # rows generated on the member's device (8192 shards x 65 rows = 70 GB in total)
                dev = torch.device("cuda", devices[m])

                def gen(n_rows, seed):
                    g = torch.Generator(device=dev)
                    g.manual_seed(seed)
                    return torch.randint(-(2**63), 2**63 - 1, (n_rows, 16, 1024), dtype=torch.int64, device=dev, generator=g)

                ta, tb, tf = gen(ns * n_a, 7100 + 3 * m), gen(ns * n_b, 7101 + 3 * m), gen(ns, 7102 + 3 * m)
                torch.cuda.synchronize(dev)
                A, B, F = ctx.upload_dense_device(ta.data_ptr(), ns * n_a), ctx.upload_dense_device(tb.data_ptr(), ns * n_b), ctx.upload_dense_device(tf.data_ptr(), ns)
                chk = min(ns, 8)  # spot check of one cell on the member's first shards (all shards: the library's own tests)
                wa, wb, wf = (t[: chk * k].cpu().numpy().view(np.uint64) for t, k in ((ta, n_a), (tb, n_b), (tf, 1)))
                del ta, tb, tf
                torch.cuda.empty_cache()
            else:
                wa, wb, wf = D.dense_rows(ns * n_a, 0.5, 7100 + 3 * m), D.dense_rows(ns * n_b, 0.5, 7101 + 3 * m), D.dense_rows(ns, 0.5, 7102 + 3 * m)
                A, B, F = ctx.upload_dense(wa), ctx.upload_dense(wb), ctx.upload_dense(wf)
                chk = ns
            keep += [A, B, F]
            per_member.append(dict(a=A, rows_a=np.arange(ns * n_a).reshape(ns, n_a), b=B, rows_b=np.arange(ns * n_b).reshape(ns, n_b), filt=F, rows_f=np.arange(ns)))
            if strong:  # the member's first shards as a query of their own, against numpy
                sub = ctx.count_matrix(A, np.arange(chk * n_a).reshape(chk, n_a), B, np.arange(chk * n_b).reshape(chk, n_b), F, np.arange(chk))
                assert int(sub[1, 2]) == int(sum(np.bitwise_count(wa[s * n_a + 1] & wb[s * n_b + 2] & wf[s]).sum() for s in range(chk))), ("member", m)
            else:
                for s in range(ns):
                    exp[1, 2] += np.bitwise_count(wa[s * n_a + 1] & wb[s * n_b + 2] & wf[s]).sum()
        mm, ref = {}, None
        for name, mode in modes:
            try:
                grp.set_reduce(mode)
            except L.FbkError as e:
                mm[name] = {"error": str(e)}
                continue
            for _ in range(3):
                tot = grp.count_matrix(per_member, n_a, n_b)
            if not strong:
                assert int(tot[1, 2]) == int(exp[1, 2]), (name, int(tot[1, 2]), int(exp[1, 2]))
            if ref is None:
                ref = tot  # every reduce mode must deliver the same matrix
            assert (tot == ref).all(), name
            it = max(5, args.steps // 20)
            lat = []
            for _ in range(it):
                t1 = time.perf_counter()
                grp.count_matrix(per_member, n_a, n_b)
                lat.append(time.perf_counter() - t1)
            lat.sort()
            dt = sum(lat) / it
            mm[name] = {"ms_per_call": dt * 1e3, "ms_per_call_median": lat[len(lat) // 2] * 1e3, "container_pairs_per_s": sum(shards_of) * 16 * n_a * n_b / dt}
            out["count_matrix"] = {"rows_per_side": args.matrix, "shards_per_member": shards_of, "scaling": "strong" if strong else "weak",
                                   "op": "fbk_group_count_matrix: GroupBy IntersectionCount matrix over every member's shards, reduced matrix on the host after every call", "modes": mm}
            print(json.dumps(out), flush=True)
        for b in keep:
            b.free()
    grp.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
