#!/usr/bin/env python3
"""Exercise the kernels that had no rocprof line in round 1 at sizes where their
launches are measurable, so that `rocprofv3 --kernel-trace --stats -- python scripts/profile_misc.py`
yields a per-kernel summary (committed under profiles/).  Prints one JSON object: per call the
algorithmic bytes (encoded payload read once + bytes written) for the roofline column of DESIGN.md.

Kernels reached: k_setop<OP> (generic pairs), k_encode_plan / k_scan_blocks / k_exclusive_scan /
k_encode_write (optimize), k_count_range, k_fold_n<AND>, k_fold_scatter<XOR/ANDNOT>, k_shift, k_flip,
k_bsi_add, k_bsi_values (+ hipcub sort), k_bsi_minmax_slot, k_bsi_range_slot / k_bsi_sum_slot, k_rows_flags, k_wire_copy (roaring
upload + download), k_validate_recount, k_recount, k_rows_vs_filter + k_topn_filter,
k_counts_to_bsi / k_cell_stats, k_count_matrix_fusedq, k_count_matrix<4>."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

N3 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, groups, filt = D.config3_flat(N3, mp="fork")
import torch  # noqa: E402

from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
fidx = np.arange(N3)
nbytes = rows.bytes + filt.bytes
out = {"config3_shards": N3, "config3_encoded_bytes": nbytes, "calls": {}}
REP = 5


def rec(name, alg_bytes, fn):
    for _ in range(REP):
        r = fn()
        if isinstance(r, tuple) and hasattr(r[0], "free"):
            r[0].free()
        elif hasattr(r, "free"):
            r.free()
    out["calls"][name] = {"algorithmic_bytes": int(alg_bytes), "launches": REP}


n_pairs = N3 * 32
pa, pb = groups[:, :32].reshape(-1), groups[:, 32:].reshape(-1)
pair_bytes = nbytes  # rows 0..31 vs rows 32..63 of every shard: every container of the rows once (filter not read)
for op, nm in ((L.OP_AND, "intersect"), (L.OP_OR, "union"), (L.OP_XOR, "xor"), (L.OP_ANDNOT, "difference")):
    rec(f"fbk_setop {nm} (k_setop), {n_pairs} mixed row pairs -> 8 KiB cells", pair_bytes + n_pairs * 16 * 8192, lambda op=op: ctx.setop(op, batch, pa, batch, pb))
rec(f"fbk_setop intersect + optimize (k_setop + k_encode_*), {n_pairs} pairs", pair_bytes, lambda: ctx.setop(L.OP_AND, batch, pa, batch, pb, L.SETOP_OPTIMIZE))
rec(f"fbk_setop union + optimize (k_setop + k_encode_*), {n_pairs} pairs", pair_bytes, lambda: ctx.setop(L.OP_OR, batch, pa, batch, pb, L.SETOP_OPTIMIZE))
rec(f"fbk_intersection_count (k_icount), {n_pairs} mixed row pairs", pair_bytes, lambda: ctx.intersection_count(batch, pa, batch, pb))
allrows = groups.reshape(-1)
rec(f"fbk_count_range [70000, 900000) (k_count_range), {allrows.size} rows", allrows.size * 2 * 2048, lambda: ctx.count_range(batch, allrows, 70000, 900000))
rec("fbk_fold_n AND of 8 rows (k_fold_n<AND>)", nbytes / 8, lambda: ctx.fold_n(L.OP_AND, batch, groups[:, :8]))
rec("fbk_fold_n XOR of 64 rows (k_fold_scatter<XOR>)", nbytes, lambda: ctx.fold_n(L.OP_XOR, batch, groups))
rec("fbk_fold_n ANDNOT of 64 rows (k_fold_scatter<ANDNOT>)", nbytes, lambda: ctx.fold_n(L.OP_ANDNOT, batch, groups))
rec("fbk_fold_n_intersection_count OR + filter (k_fold_scatter<OR>)", nbytes, lambda: ctx.union_n_intersection_count(batch, groups, F, fidx))
rec(f"fbk_shift (k_shift), {allrows.size} rows", nbytes + allrows.size * 16 * 8192, lambda: ctx.shift(batch, allrows))
rec(f"fbk_flip [123, 900000] (k_flip), {allrows.size} rows", nbytes + allrows.size * 16 * 8192, lambda: ctx.flip(batch, allrows, 123, 900000))
rec(f"fbk_rows column filter (k_rows_flags + scan / select), {allrows.size} rows", allrows.size * 16 * 16, lambda: ctx.rows(batch, allrows, (3 << 16) + 77))
rec("fbk_topn MinThreshold + Tanimoto (k_rows_vs_filter, k_row_cardinality, k_topn_filter)", nbytes, lambda: ctx.topn(batch, groups, 10, F, fidx, tanimoto_threshold=2))
rec("fbk_topk_bsi (k_rows_vs_filter, k_counts_to_bsi, k_cell_stats)", nbytes, lambda: ctx.topk_bsi(batch, groups, F, fidx))
rec("fbk_count_matrix 32 x 32 + filter, in-kernel decode (k_fused_program + k_count_matrix_fusedq)", nbytes, lambda: ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, fidx))
ctx.set_option("matrix_fused", 0)
rec("fbk_count_matrix 8 x 8 + filter, generic pair kernel (k_count_matrix<4>)", nbytes * 17 / 65, lambda: ctx.count_matrix(batch, groups[:, :8], batch, groups[:, 32:40], F, fidx))
ctx.set_option("matrix_fused", -1)
# serialised roaring: download the union result and upload it again
u, _ = ctx.union_n(batch, groups, L.SETOP_OPTIMIZE)
blob = u.to_roaring()
rec(f"fbk_batch_download_roaring (k_wire_copy), {len(blob)} bytes", 2 * len(blob), lambda: u.to_roaring())
rec(f"fbk_batch_upload_roaring (k_wire_copy + k_validate_recount), {len(blob)} bytes", 2 * len(blob), lambda: ctx.upload_roaring(blob))
u.free()
# BSI: 96 shards x (64 planes + exists + sign), dense
n5, depth = 96, 64
g = torch.Generator(device="cuda")
g.manual_seed(5)
w = torch.randint(-(2**63), 2**63 - 1, (n5 * (depth + 2), 16, 1024), dtype=torch.int64, device="cuda", generator=g).cpu().numpy().view(np.uint64)
w = w.reshape(n5, depth + 2, 16, 1024)
w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
bsi = ctx.upload_dense(w.reshape(-1))
base = np.arange(n5, dtype=np.uint32) * (depth + 2)
pl = n5 * 16 * 8192
rec("fbk_bsi_range GT 2^62 (k_bsi_range_slot)", pl * (depth + 3), lambda: ctx.bsi_range(bsi, base, L.BSI_GT, depth, 1 << 62))
rec("fbk_bsi_sum (k_bsi_sum_slot)", pl * (depth + 2), lambda: ctx.bsi_sum(bsi, base, depth))
rec("fbk_bsi_range_sum GT 2^62, one pass (k_bsi_range_sum_half)", pl * (depth + 2), lambda: ctx.bsi_range_sum(bsi, base, L.BSI_GT, depth, 1 << 62))
rec("fbk_bsi_range_between_sum 2^60 .. 2^62, one pass (k_bsi_between_sum_half)", pl * (depth + 2), lambda: ctx.bsi_range_between_sum(bsi, base, depth, 1 << 60, 1 << 62))
rec("fbk_bsi_min (k_bsi_minmax_slot)", pl * (depth + 2), lambda: ctx.bsi_min(bsi, base, depth))
rec("fbk_bsi_max (k_bsi_minmax_slot)", pl * (depth + 2), lambda: ctx.bsi_max(bsi, base, depth))
px = np.arange(n5 * 16, dtype=np.uint32).reshape(n5, 16) + 2  # 16 planes of each shard as one unsigned operand (rows base + 2 ..)
px = (base[:, None] + 2 + np.arange(16)[None, :]).astype(np.uint32)
py = (base[:, None] + 18 + np.arange(16)[None, :]).astype(np.uint32)
rec("fbk_bsi_add 16 + 16 planes (k_bsi_add)", pl * (32 + 17), lambda: ctx.bsi_add(bsi, px, bsi, py))
small = ctx.upload_dense(w[:4, :18].reshape(-1))  # Distinct: 4 shards x (16 planes + exists + sign): 4 M values
rec("fbk_bsi_distinct 4 shards x 16 planes (k_bsi_values + radix sort + unique)", 4 * 16 * 8192 * 18 + 4 * (1 << 20) * 8, lambda: ctx.bsi_distinct(small, np.arange(4, dtype=np.uint32) * 18, 16))
if n5 >= 32:  # the same kernel where the launch is not most of it: 32 shards = 32 M values (344 MB: 75 read + 268 written)
    mid = ctx.upload_dense(w[:32, :18].reshape(-1))
    rec("fbk_bsi_distinct 32 shards x 16 planes (k_bsi_values + radix sort + unique)", 32 * 16 * 8192 * 18 + 32 * (1 << 20) * 8, lambda: ctx.bsi_distinct(mid, np.arange(32, dtype=np.uint32) * 18, 16))
print(json.dumps(out))
