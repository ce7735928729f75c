"""Driver for rocprofv3 --pmc passes over the scatter kernels on config 3's rows: k_fold_scatter<OR>
(Union of 64 rows + IntersectionCount(filter)) and k_rows_vs_filter (64 rows against the filter row)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
fidx = np.arange(n)
for _ in range(3):
    ctx.union_n_intersection_count(batch, groups, F, fidx)
    ctx.count_matrix(batch, groups, F, fidx.reshape(n, 1))
