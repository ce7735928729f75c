#!/usr/bin/env python3
"""Where a wave of k_icount2 spends its life (option pair_stamp: waves report shader cycles of a phase instead of counts),
on config 3's row pairs, overall and per type pair of the item:
    python scripts/pairs_stamps.py [shards=64]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import _experiments  # noqa: E402

_experiments.use()  # the -DFBK_EXPERIMENTS build: the ablation / cycle-stamp options do not exist in the product library
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
ctx.set_option("pair_kernels", 2)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
pa, pb = groups[:, :32].reshape(-1), groups[:, 32:].reshape(-1)
d = rows.descs()
types = np.zeros((rows.n_rows, 16), dtype=np.uint8)
types[d["row"], d["key"] & 15] = d["type"]
plan = ctx.plan(batch, pa, batch, pb)
out = {"shards": n, "pairs": int(pa.size), "unit": "shader cycles per wave (one item per wave), mean over the items; a pair's 16 items are summed by k_sum_wave_counts, so per-class figures come from runs that skip the other classes (pair_ablate)"}
names = {1: "launch -> descriptors", 2: "descriptors -> batch 0 landed", 3: "batch 0 -> decoded (incl. further loads)", 4: "whole wave"}
for label, abl in (("all items", 0), ("array x array only (runs and bitmap x array skipped)", 8 | 32), ("run items only", 16 | 32)):
    ctx.set_option("pair_ablate", abl)
    res = {}
    for st in (1, 2, 3, 4):
        ctx.set_option("pair_stamp", st)
        for _ in range(3):
            plan.intersection_count()
        res[names[st]] = float(plan.read().sum()) / (pa.size * 16)
    out[label] = res
ctx.set_option("pair_stamp", 0)
ctx.set_option("pair_ablate", 0)
print(json.dumps(out, indent=1))
