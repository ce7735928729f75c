"""Driver for rocprofv3 --pmc passes over the row-pair kernels (config 3 rows 0..31 x rows 32..63):
    python scripts/pairs_pmc.py <shards> <options: name=value,...> [op: icount|and|or|xor|andnot]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
opts = sys.argv[2] if len(sys.argv) > 2 else ""
op = sys.argv[3] if len(sys.argv) > 3 else "icount"
rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
for kv in [x for x in opts.split(",") if x and x != "0"]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
pa, pb = groups[:, :32].reshape(-1), groups[:, 32:].reshape(-1)
plan = ctx.plan(batch, pa, batch, pb)
code = {"and": L.OP_AND, "or": L.OP_OR, "xor": L.OP_XOR, "andnot": L.OP_ANDNOT}
for _ in range(3):
    if op == "icount":
        plan.intersection_count()
    else:
        plan.setop(code[op])
ctx.synchronize()
