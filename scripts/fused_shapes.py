"""Count matrix 32 x 32 + filter on three row populations, in-kernel decode (default) against the two-kernel
densify path and the generic pair kernel: where does each win?
    python scripts/fused_shapes.py [shards=64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import datagen as D  # noqa: E402
from featurebase_amd.roaring import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def flat(kind):
    rng = D.rng_for(4242)
    fr, ff = D.FlatRows(), D.FlatRows()
    for s in range(n):
        for r in range(64):
            for slot in range(16):
                if kind == "bitmaps":
                    w = rng.integers(0, 2**64, 1024, dtype=np.uint64)
                    fr.add(s * 64 + r, s * 16 + slot, 2, w, int(np.bitwise_count(w).sum()))
                elif kind == "tiny":
                    v = np.sort(rng.choice(65536, size=int(rng.integers(1, 20)), replace=False)).astype(np.uint16)
                    fr.add(s * 64 + r, s * 16 + slot, 1, v, v.size)
                else:  # big arrays, 3000-4000 values
                    v = np.sort(rng.choice(65536, size=int(rng.integers(3000, 4000)), replace=False)).astype(np.uint16)
                    fr.add(s * 64 + r, s * 16 + slot, 1, v, v.size)
        for slot in range(16):
            w = rng.integers(0, 2**64, 1024, dtype=np.uint64)
            ff.add(s, s * 16 + slot, 2, w, int(np.bitwise_count(w).sum()))
    fr.n_rows, ff.n_rows = n * 64, n
    return fr, ff


groups = np.arange(n * 64, dtype=np.uint32).reshape(n, 64)
for kind in ("bitmaps", "big_arrays", "tiny"):
    fr, ff = flat(kind)
    batch = ctx.upload_flat(fr.descs(), fr.payload(), fr.n_rows)
    F = ctx.upload_flat(ff.descs(), ff.payload(), ff.n_rows)
    gb = lambda: ctx.count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, np.arange(n))  # noqa: E731
    ctx.set_option("matrix_fused", 1)
    ref = gb()
    tf = t(gb)
    ctx.set_option("matrix_fused", 0)
    ctx.set_option("matrix_densify", 1)
    assert (gb() == ref).all()
    td = t(gb)
    ctx.set_option("matrix_densify", 0)
    assert (gb() == ref).all()
    tg = t(gb)
    ctx.set_option("matrix_fused", -1)
    ctx.set_option("matrix_densify", -1)
    print(f"{kind:10s} {n} shards, {fr.bytes / 1e6:8.1f} MB encoded: in-kernel decode {tf:8.1f} us | densify + dense {td:8.1f} us | generic pair kernel {tg:8.1f} us")
    batch.free()
    F.free()
