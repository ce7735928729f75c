#!/usr/bin/env python3
"""Row-pair kernels on BASELINE config 3's mixed rows: rows 0..31 against rows 32..63 of every shard
(bitmap x array, array x array of very different lengths, a quarter of the containers runs) — the shape
RowSegment.IntersectionCount / Intersect / Union / Difference / Xor (row.go:556-610) see on non-dense
rows.  Times the kernel of a plan alone (HIP events around back-to-back launches on the library's stream)
for each value of option pair_kernels, checks every pair of the first variant against the oracle and the
variants against each other.

    python scripts/bench_pairs.py [--shards 64] [--iters 30] [--variants 1,2] [--out profiles/pairs_r03.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--variants", default="pair_kernels=1;pair_kernels=2", help="';'-separated option sets, each a ','-separated list of name=value")
    ap.add_argument("--out", default="")
    ap.add_argument("--only-count", action="store_true")
    ap.add_argument("--ops", default="", help="comma-separated subset of the operation names (e.g. 'intersectionCount,intersect + optimize()')")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--no-check", action="store_true", help="timing experiments on library builds that skip work: no comparison with the oracle")
    args = ap.parse_args()
    if "pair_ablate" in args.variants or "pair_stamp" in args.variants or "pair_spw" in args.variants or any("ablate" in o or "stamp" in o for o in args.opt):
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import _experiments

        if not os.environ.get("FBK_LIB_PATH"):  # (an A/B run names its own -DFBK_EXPERIMENTS build)
            _experiments.use()  # those options exist in the -DFBK_EXPERIMENTS build only
    rows, groups, filt = D.config3_flat(args.shards, mp="fork")
    import torch

    from featurebase_amd import lib as L
    from featurebase_amd.roaring import Context
    from oracle import pybatch as PB

    ctx = Context(0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    OA = PB.RowSet.from_flat(rows.descs(), rows.payload(), rows.n_rows)
    pa, pb = groups[:, :32].reshape(-1), groups[:, 32:].reshape(-1)
    n_pairs = pa.size
    d = rows.descs()
    types = np.zeros((rows.n_rows, 16), dtype=np.uint8)
    types[d["row"], d["key"] & 15] = d["type"]
    ta, tb = types[pa], types[pb]
    mix = {f"{'nabr'[x]}x{'nabr'[y]}": int(((ta == x) & (tb == y)).sum()) for x in range(4) for y in range(4) if ((ta == x) & (tb == y)).any()}
    exp = PB.intersection_count(OA, pa, OA, pb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {"shards": args.shards, "pairs": int(n_pairs), "encoded_bytes": int(rows.bytes), "type_pairs (n nil, a array, b bitmap, r run)": mix, "variants": {}}
    ops = [("intersectionCount", None), ("intersect", L.OP_AND), ("union", L.OP_OR), ("difference", L.OP_ANDNOT), ("xor", L.OP_XOR),
           ("intersect + optimize()", (L.OP_AND, L.SETOP_OPTIMIZE)), ("difference + optimize()", (L.OP_ANDNOT, L.SETOP_OPTIMIZE)),
           ("union + optimize()", (L.OP_OR, L.SETOP_OPTIMIZE)), ("xor + optimize()", (L.OP_XOR, L.SETOP_OPTIMIZE))]
    if args.only_count:
        ops = ops[:1]
    if args.ops:
        ops = [o for o in ops if o[0] in args.ops.split(",")]
    ref_counts = {}
    for var in args.variants.split(";"):
        for kv in var.split(","):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        plan = ctx.plan(batch, pa, batch, pb)
        out = {}
        for name, op in ops:
            flags = 0
            if isinstance(op, tuple):
                op, flags = op
            fn = plan.intersection_count if op is None else (lambda o=op, f=flags: plan.setop(o, f))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            samples = []
            for _ in range(5):
                e0.record(st)
                for _ in range(args.iters):
                    fn()
                e1.record(st)
                torch.cuda.synchronize()
                samples.append(e0.elapsed_time(e1) * 1e3 / args.iters)
            us = sorted(samples)[len(samples) // 2]
            counts = plan.read()
            ablated = args.no_check or ("pair_ablate" in var and "pair_ablate=0" not in var)
            if ablated:
                pass  # timing experiment: results are wrong by construction
            elif op is None:
                assert (counts == exp).all(), f"pair_kernels={var}: intersectionCount differs from the oracle"
            else:
                name_ref = name.split(" +")[0]
                if name_ref not in ref_counts:
                    _, ecnt = PB.setop({L.OP_AND: PB.OP_AND, L.OP_OR: PB.OP_OR, L.OP_XOR: PB.OP_XOR, L.OP_ANDNOT: PB.OP_ANDNOT}[op], OA, pa, OA, pb)
                    ref_counts[name_ref] = ecnt
                assert ablated or (counts == ref_counts[name_ref]).all(), f"pair_kernels={var}: {name} cardinalities differ from the oracle"
            nbytes = rows.bytes + (0 if op is None else plan.output().info()[2] if flags else n_pairs * 16 * 8192)
            out[name] = {"us": us, "min_us": min(samples), "algorithmic_bytes": int(nbytes), "TBps": nbytes / us / 1e6, "frac_of_8TBps": nbytes / us / 1e6 / 8.0,
                         "pairs_per_s": n_pairs * 16 / (us * 1e-6)}
        plan.free()
        res["variants"][var] = out
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
