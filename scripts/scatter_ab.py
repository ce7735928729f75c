"""The scatter kernels on config 3's rows (256 shards): Union-of-64 then IntersectionCount (k_fold_scatter<OR>), the
TopN / TopK shape (k_rows_vs_filter) and the materialised Union-of-64 + optimize(), kernel time by the library's HIP events.
Run once per library build (FBK_LIB_PATH) for an A/B of two builds on the same box:

    python scripts/scatter_ab.py [shards=256] [runs=30]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
fidx = np.arange(n)
nbytes = rows.bytes + filt.bytes
ctx.set_option("time_kernels", 1)
qs = {"fold_icount (k_fold_scatter<OR>)": ctx.prepare_fold_intersection_count(L.OP_OR, batch, groups, F, fidx),
      "topk shape (k_rows_vs_filter)": ctx.prepare_count_matrix(batch, groups, F, fidx.reshape(-1, 1)),
      "union materialised + optimize (k_fold_scatter<OR, optimize>)": ctx.prepare_fold(L.OP_OR, batch, groups, L.SETOP_OPTIMIZE) if hasattr(ctx, "prepare_fold") else None}
out = {"lib": os.environ.get("FBK_LIB_PATH", "product"), "shards": n, "bytes": nbytes, "kernel_us": {}}
chk = {}
for name, q in qs.items():
    if q is None:
        continue
    ts = []
    for r in range(runs + 3):
        q.run()
        ctx.synchronize()
        if r >= 3:
            ts.append(ctx.get_option("last_kernel_ns") / 1e3)
    ts.sort()
    res = q.read()
    chk[name] = int(np.asarray(res[0] if isinstance(res, tuple) else res).astype(np.uint64).sum())
    out["kernel_us"][name] = {"median": ts[len(ts) // 2], "min": ts[0], "p90": ts[(len(ts) * 9) // 10], "frac_of_8TBps": nbytes / (ts[len(ts) // 2] * 1e-6) / 8e12}
out["checksums"] = chk
print(json.dumps(out))
