"""Cycle stamps of one block of k_count_matrix_fusedq (option matrix_fused_ablate = 32: the instrumented
build prints them to stderr; that build SPILLS, so the stamps over-state the consumers' start — see
profiles/r05_fused_cycle_stamps_config4_prof_build_spills.txt).   python scripts/fused_prof.py [shards=256] [ablate bits=0] [unused] [config=3]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ab = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 3  # 3: config 3's rank-law rows; 4: config 4's log-uniform rows
if cfg == 4:
    rows, ga, gb, filt, _ = D.config4_flat(n, mp="fork")
    groups = np.concatenate([ga, gb], axis=1)
else:
    rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
ctx.set_option("matrix_fused", 1)
print(f"[fused prof] config {cfg}", file=sys.stderr)
for _ in range(2):
    ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, np.arange(n))
ctx.set_option("matrix_fused_ablate", 32 | ab)
ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, np.arange(n))
