"""k_count_matrix_fused on config 3's rows (GroupBy 32 x 32 + filter, 256 shards): kernel time by slots per block (option
matrix_spb; 1 block per CU at a time, so 512 blocks are two rounds), heavy-row shadows on and off, interleaved in one process.

    python scripts/fused_spb_sweep.py [shards=256] [rounds=5]
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
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows, groups, filt = D.config3_flat(n, mp="fork")
from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
fidx = np.arange(n)
ctx.set_option("time_kernels", 1)
out = {"shards": n, "bytes": rows.bytes + filt.bytes, "kernel_us": {}}
ref = None
for shadow in (1, 0):
    ctx.set_option("matrix_shadow", shadow)
    batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)  # (the shadow decision is per batch: a fresh one per mode)
    F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
    q = ctx.prepare_count_matrix(batch, groups[:, :32], batch, groups[:, 32:], F, fidx)
    times = {}
    for r in range(rounds + 1):
        for spb in (0, 16, 8, 4, 2):
            ctx.set_option("matrix_spb", spb)
            q.run()
            ctx.synchronize()
            if r == 0:
                tot = q.read()
                ref = tot if ref is None else ref
                assert (tot == ref).all(), (shadow, spb)
            else:
                times.setdefault(spb, []).append(ctx.get_option("last_kernel_ns") / 1e3)
    ctx.set_option("matrix_spb", 0)
    for spb, t in times.items():
        t.sort()
        out["kernel_us"][f"shadows={shadow} spb={'heuristic' if spb == 0 else spb}"] = {"median": t[len(t) // 2], "min": t[0], "max": t[-1]}
    q.free()
    batch.free()
    F.free()
print(json.dumps(out, indent=1))
