"""Timing experiments on k_count_matrix_fused: config 3's mixed rows, GroupBy 32 x 32 + filter,
with parts of the kernel switched off (option matrix_fused_ablate; the counts are wrong then)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows, groups, filt = D.config3_flat(n, mp="fork")
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
fidx = np.arange(n)
nbytes = rows.bytes + filt.bytes


def t(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


gb = lambda: ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, fidx)  # noqa: E731
print(f"{n} shards, {nbytes/1e6:.1f} MB encoded")
ref = gb()  # default path: densify + dense matrix-core kernel
ctx.set_option("matrix_fused", 1)
assert (gb() == ref).all()
for spb in (0, 16, 8, 4, 2, 1):
    ctx.set_option("matrix_spb", spb)
    print(f"spb={spb:2d} fused {t(gb):8.1f} us")
ctx.set_option("matrix_spb", 0)
for ab, what in [(0, "full"), (1, "no consumer math"), (2, "no arrays"), (4, "no runs"), (8, "no bitmap DMA"), (16, "no zeroing"), (6, "no arrays, no runs"), (14, "no decode at all (zero only)"),
                 (30, "producers idle"), (31, "everything off (barriers only)"), (7, "no math, no arrays, no runs")]:
    ctx.set_option("matrix_fused_ablate", ab)
    print(f"ablate={ab:2d} {what:36s} {t(gb):8.1f} us")
for ab in (32, 33):
    ctx.set_option("matrix_fused_ablate", ab)
    sys.stderr.flush()
    gb()
    gb()
ctx.set_option("matrix_fused_ablate", 0)
assert (gb() == ref).all()
ctx.set_option("matrix_fused", 0)
print(f"densify + dense kernel {t(gb):8.1f} us")
assert (gb() == ref).all()
