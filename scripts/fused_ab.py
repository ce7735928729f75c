"""The count matrix over encoded rows (k_fused_program + k_count_matrix_fusedq) in ONE process on ONE box: config 3's rows
(GroupBy 32 x 32 + filter, rank-law densities with runs) and config 4 as SURVEY 8d writes it (log-uniform densities), with and
without the heavy-row shadows (option matrix_shadow), with and without the filter, and by container slots per block (option
matrix_spb).  Prepared queries, kernel time from the library's events (option time_kernels), every variant checked against
the first.  (Round 5 used this script for the A/B of the program-driven kernels against round 4's kernel — option
matrix_fused_program, since removed with the kernels that lost: profiles/r05_fused_program_ab.json,
r05_fused_ab_specialised_producers.json, r05_fused_ab_final_kernel.json.)

    python scripts/fused_ab.py [shards3=256] [shards4=1024] > gpurun_out/.../fused_ab.json
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
n4 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sets = []
if n3:
    r3, g3, f3 = D.config3_flat(n3, mp="fork")
    sets.append(("config3", r3.descs(), r3.payload(), r3.n_rows, g3, f3.descs(), f3.payload(), n3, r3.bytes + f3.bytes))
if n4:
    r4, ga4, gb4, f4, _ = D.config4_flat(n4, mp="fork")
    g4 = np.concatenate([ga4, gb4], axis=1)
    sets.append(("config4_loguniform", r4.descs(), r4.payload(), r4.n_rows, g4, f4.descs(), f4.payload(), n4, r4.bytes + f4.bytes))
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
out = {"note": "kernel_us: median / min of 12 runs of the prepared query's dominant kernel (library events); frac on the ENCODED bytes at 8 TB/s", "sets": {}}
for name, d, p, nr, g, fd, fp, n, nbytes in sets:
    res = []
    ref = None
    fidx = np.arange(n)
    for shadow, thr, apref in ((1, 2048, 2), (0, 2048, 2)):
        ctx.set_option("matrix_shadow", shadow)
        ctx.set_option("matrix_shadow_array", thr)
        batch = ctx.upload_flat(d, p, nr)  # (a shadow is built once per batch, with the options in force then)
        F = ctx.upload_flat(fd, fp, n)
        for rep in (0, 1):  # (two rounds of the same measurement: the first launches after an upload run slower)
            q = ctx.prepare_count_matrix(batch, g[:, :32], batch, g[:, 32:], F, fidx)
            q.run()
            got = q.read()
            if ref is None:
                ref = got
            ok = bool((got == ref).all())
            ctx.set_option("time_kernels", 1)
            ts = []
            for _ in range(12):
                q.run()
                torch.cuda.synchronize()
                ts.append(ctx.get_option("last_kernel_ns") / 1e3)
            ctx.set_option("time_kernels", 0)
            ts.sort()
            # without the filter
            qn = ctx.prepare_count_matrix(batch, g[:, :32], batch, g[:, 32:])
            qn.run()
            ctx.set_option("time_kernels", 1)
            tn = []
            for _ in range(6):
                qn.run()
                torch.cuda.synchronize()
                tn.append(ctx.get_option("last_kernel_ns") / 1e3)
            ctx.set_option("time_kernels", 0)
            tn.sort()
            res.append({"shadow": shadow, "shadow_array": thr, "round": rep, "kernel_us": ts[len(ts) // 2], "kernel_us_min": ts[0], "same_counts": ok,
                        "frac": nbytes / (ts[len(ts) // 2] * 1e-6) / 8e12, "no_filter_kernel_us": tn[len(tn) // 2]})
            print(name, res[-1], file=sys.stderr, flush=True)
            q.free()
            qn.free()
        if shadow == 1:  # slots per block of the default kernel (16: one block per shard; fewer: more, shorter blocks)
            sweep = []
            for spb in (16, 8, 4):
                ctx.set_option("matrix_spb", spb)
                q = ctx.prepare_count_matrix(batch, g[:, :32], batch, g[:, 32:], F, fidx)
                q.run()
                ok = bool((q.read() == ref).all())
                ctx.set_option("time_kernels", 1)
                ts = []
                for _ in range(10):
                    q.run()
                    torch.cuda.synchronize()
                    ts.append(ctx.get_option("last_kernel_ns") / 1e3)
                ctx.set_option("time_kernels", 0)
                ts.sort()
                sweep.append({"spb": spb, "kernel_us": ts[len(ts) // 2], "kernel_us_min": ts[0], "same_counts": ok})
                print(name, "spb", sweep[-1], file=sys.stderr, flush=True)
                q.free()
            ctx.set_option("matrix_spb", 0)
            out.setdefault("spb_sweep", {})[name] = sweep
        batch.free()
        F.free()
    out["sets"][name] = {"shards": n, "encoded_bytes": int(nbytes), "variants": res}
print(json.dumps(out))
