"""k_count_matrix_fused on config 3's mixed rows (GroupBy 32 x 32 + filter): parity against the
densify path, time per launch, parts of the kernel switched off (option matrix_fused_ablate; the
counts are wrong then), and the first version of the kernel / the two-kernel path beside it.

    python scripts/fused_bench.py [shards=256]
"""
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
print(f"{n} shards, {nbytes/1e6:.1f} MB encoded (one-shot call: ~45 us of row upload / result download around the kernel)")
ctx.set_option("matrix_fused", 0)
ref = gb()
print(f"densify + dense kernel      {t(gb):8.1f} us")
ctx.set_option("matrix_fused", 1)
got = gb()
ok2 = bool((got == ref).all())
print(f"fused                      {t(gb):8.1f} us  parity {ok2}  ({nbytes / t(gb) / 1e6:.2f} TB/s)")



def kernel_us(fn, iters=12):
    ctx.set_option("time_kernels", 1)
    ts = []
    for _ in range(iters):
        fn()
        ts.append(ctx.get_option("last_kernel_ns") / 1e3)
    ctx.set_option("time_kernels", 0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


# heavy-row shadows (option matrix_shadow): every variant on its own upload of the rows (a shadow is built once per batch)
for shadow, thr, apref in ((0, 0, 1), (1, 4096, 1), (1, 2048, 1), (1, 2048, 2), (1, 1024, 1), (1, 1024, 2), (1, 512, 1), (0, 0, 1)):
    ctx.set_option("matrix_shadow", shadow)
    ctx.set_option("matrix_shadow_array", thr)
    ctx.set_option("matrix_shadow_apref", apref)
    bt = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    fn = lambda: ctx.count_matrix(bt, groups[:, :32], bt, groups[:, 32:], F, fidx)  # noqa: E731
    ok = bool((fn() == ref).all())
    med, lo = kernel_us(fn)
    print(f"fused, shadows {'arrays > %4d + runs, %d item(s) ahead' % (thr, apref) if shadow else 'off                                 '}  kernel {med:7.1f} us (min {lo:6.1f})  parity {ok}")
    bt.free()
ctx.set_option("matrix_shadow", 1)
ctx.set_option("matrix_shadow_array", 2048)
ctx.set_option("matrix_shadow_apref", 2)
if not ok2:
    bad = np.argwhere(got != ref)
    print("mismatches:", len(bad), bad[:10].tolist(), got[got != ref][:10].tolist(), ref[got != ref][:10].tolist())
quick = os.environ.get("FUSED_QUICK") == "1"  # the fused line and the 'no runs' / 'no arrays' ablations only
for spb in () if quick else (16, 8, 4, 2, 1):
    ctx.set_option("matrix_spb", spb)
    print(f"fused spb={spb:2d}               {t(gb):8.1f} us")
ctx.set_option("matrix_spb", 0)
for ab, what in [(1, "no consumer math"), (2, "no arrays"), (4, "no runs"), (8, "no bitmap rows"), (6, "no arrays, no runs"), (14, "no decode at all"),
                 (15, "barriers + work lists only"), (3, "no math, no arrays"), (5, "no math, no runs")]:
    if quick and ab not in (2, 4):
        continue
    ctx.set_option("matrix_fused_ablate", ab)
    print(f"fused ablate={ab:2d} {what:28s} {t(gb):8.1f} us")
ctx.set_option("matrix_fused_ablate", 0)
assert (gb() == ref).all() or not ok2
