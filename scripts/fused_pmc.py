"""Driver for rocprofv3 --pmc passes over k_count_matrix_fused (config 3 rows, GroupBy 32 x 32 + filter)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ablate = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if ablate:  # parts of the kernel switched off: the -DFBK_EXPERIMENTS build (the option does not exist in the product library)
    import _experiments  # noqa: E402

    _experiments.use()
rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
if ablate:
    ctx.set_option("matrix_fused_ablate", ablate)
for _ in range(3):
    ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, np.arange(n))
