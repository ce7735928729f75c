"""Small GroupBy shapes on config 3's mixed rows: which count-matrix path serves an n_a x n_b (+ filter) query of a few
rows per side best — the generic pair kernel (k_count_matrix<4>), the in-kernel-decode matrix-core kernel
(k_count_matrix_fusedq) — and fbk_count_range.  Interleaved A/B inside one process, kernel
time by the library's own HIP events (option time_kernels) and the prepared query's GPU time.

    python scripts/small_shapes_ab.py [shards=64] [rounds=7]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows, groups, filt = D.config3_flat(n, mp="fork")
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
fidx = np.arange(n)
ctx.set_option("time_kernels", 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {"shards": n, "shapes": {}}
# (round 4 also ran "densify + dense" — k_densify_rows into temporary rows, then the dense kernel; that path was removed in round 5,
# its numbers are in profiles/r04_small_shapes_ab.json)
VARIANTS = {"generic (k_count_matrix<4>)": {"matrix_fused": 0}, "fused (k_count_matrix_fusedq)": {"matrix_fused": 1}, "library default": {"matrix_fused": -1}}
for (na, nb, use_f) in ((8, 8, True), (8, 8, False), (4, 16, True), (16, 16, True), (2, 2, True), (3, 30, False)):
    ra, rb = groups[:, :na], groups[:, 32:32 + nb]
    q = ctx.prepare_count_matrix(batch, ra, batch, rb, F if use_f else None, fidx if use_f else None)
    ref, res = None, {}
    for r in range(rounds + 1):
        for name, opts in VARIANTS.items():
            for k, v in opts.items():
                ctx.set_option(k, v)
            e0.record(st)
            q.run()
            e1.record(st)
            ctx.synchronize()
            torch.cuda.synchronize()
            if r == 0:
                tot = q.read()
                ref = tot if ref is None else ref
                assert (tot == ref).all(), (na, nb, name)
            else:
                res.setdefault(name, []).append((e0.elapsed_time(e1) * 1e3, ctx.get_option("last_kernel_ns") / 1e3))
    ctx.set_option("matrix_fused", -1)
    q.free()
    out["shapes"][f"{na} x {nb}" + (" + filter" if use_f else "")] = {
        k: {"gpu_us": sorted(x[0] for x in v)[len(v) // 2], "kernel_us": sorted(x[1] for x in v)[len(v) // 2]} for k, v in res.items()}
allrows = groups.reshape(-1)
ts = []
for r in range(rounds + 1):
    e0.record(st)
    c = ctx.count_range(batch, allrows, 70000, 900000)
    e1.record(st)
    torch.cuda.synchronize()
    if r:
        ts.append(e0.elapsed_time(e1) * 1e3)
out["count_range"] = {"rows": int(allrows.size), "call_us": sorted(ts)[len(ts) // 2]}
print(json.dumps(out, indent=1))
