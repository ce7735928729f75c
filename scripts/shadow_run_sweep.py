#!/usr/bin/env python3
"""Which run containers should the count matrix over encoded rows read through an 8 KiB dense shadow, and which decode in place?
Option matrix_shadow_run (run containers of MORE than this many runs are shadowed; 0 = all of them, rounds 3-5) swept on config
3's rows (run containers of 16 / 32 / 128 / 1024 runs, a quarter of all containers) in ONE process: the prepared 32 x 32 + filter
query, kernel time from the library's events, every variant's matrix compared with the first, shadow bytes from fbk_batch_memory.

    python scripts/shadow_run_sweep.py [shards=256] > gpurun_out/.../shadow_run_sweep.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r3, g3, f3 = D.config3_flat(n3, mp="fork")
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
d, p, nr, fd, fp = r3.descs(), r3.payload(), r3.n_rows, f3.descs(), f3.payload()
nbytes = r3.bytes + f3.bytes
fidx = np.arange(n3)
out = {"shards": n3, "encoded_bytes": int(nbytes), "variants": []}
ref = None
for rep in range(2):
    for thr in (0, 16, 32, 128, 1024, 65536):
        ctx.set_option("matrix_shadow_run", thr)
        batch = ctx.upload_flat(d, p, nr)  # (a shadow is built once per batch, with the options in force then)
        F = ctx.upload_flat(fd, fp, n3)
        q = ctx.prepare_count_matrix(batch, g3[:, :32], batch, g3[:, 32:], F, fidx)
        q.run()
        got = q.read()
        if ref is None:
            ref = got
        ok = bool((got == ref).all())
        for _ in range(20):  # past the clock dip of the first launches
            q.run()
        torch.cuda.synchronize()
        ctx.set_option("time_kernels", 1)
        ts = []
        for _ in range(16):
            q.run()
            torch.cuda.synchronize()
            ts.append(ctx.get_option("last_kernel_ns") / 1e3)
        ctx.set_option("time_kernels", 0)
        ts.sort()
        try:
            mem = batch.memory()
        except Exception as e:  # noqa: BLE001
            mem = str(e)
        out["variants"].append({"rep": rep, "matrix_shadow_run": thr, "kernel_us_median": round(ts[len(ts) // 2], 1), "kernel_us_min": round(ts[0], 1),
                                "frac_of_8TBps_on_encoded_bytes": round(nbytes / (ts[len(ts) // 2] * 1e-6) / 8e12, 4), "equal_to_first": ok, "memory": mem})
        print(json.dumps(out["variants"][-1]), file=sys.stderr, flush=True)
        q.free()
        batch.free()
        F.free()
ctx.set_option("matrix_shadow_run", 0)
print(json.dumps(out, indent=1))
