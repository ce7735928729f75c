"""Units by ticket against units by block id (option matrix_tickets, round 6) for the count matrices, ONE process on ONE box,
variants alternating: the encoded-row kernel (k_count_matrix_fusedq) on config 3's rows (256 shards) and on config 4 as SURVEY
8d writes it (log-uniform densities, 1024 shards), the dense kernel (k_count_matrix_mfma) on 1024 shards of dense rows — each by
container slots per block of the first tier (option matrix_spb; 0 = the library's choice).  Prepared queries, kernel time from
the library's events (option time_kernels), every variant's counts checked against the first.

    python scripts/tickets_ab.py [shards3=256] [shards4=1024] [dense=1024] > gpurun_out/.../tickets_ab.json
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
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
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
out = {"note": "kernel_us: median / min of the prepared query's dominant kernel over `runs` runs per round, rounds of all variants alternating (library events); "
               "frac on the encoded bytes at 8 TB/s", "sets": {}}


def time_variants(name, make_query, nbytes, variants, rounds=3, runs=8):
    ref = None
    acc = {v: [] for v in variants}
    same = {v: True for v in variants}
    for rnd in range(rounds):
        for v in variants:
            tickets, spb = v
            ctx.set_option("matrix_tickets", tickets)
            ctx.set_option("matrix_spb", spb)
            q = make_query()
            q.run()
            got = q.read()
            if ref is None:
                ref = got
            same[v] = same[v] and bool((got == ref).all())
            ctx.set_option("time_kernels", 1)
            for _ in range(runs):
                q.run()
                torch.cuda.synchronize()
                acc[v].append(ctx.get_option("last_kernel_ns") / 1e3)
            ctx.set_option("time_kernels", 0)
            q.free()
    ctx.set_option("matrix_tickets", 1)
    ctx.set_option("matrix_spb", 0)
    res = []
    for v in variants:
        ts = sorted(acc[v])
        res.append({"matrix_tickets": v[0], "matrix_spb": v[1], "kernel_us": round(ts[len(ts) // 2], 1), "kernel_us_min": round(ts[0], 1), "n": len(ts), "same_counts": same[v],
                    "frac": round(nbytes / (ts[len(ts) // 2] * 1e-6) / 8e12, 4)})
        print(name, res[-1], file=sys.stderr, flush=True)
    return res


for name, d, p, nr, g, fd, fp, n, nbytes in sets:
    fidx = np.arange(n)
    batch = ctx.upload_flat(d, p, nr)
    F = ctx.upload_flat(fd, fp, n)
    variants = [(0, 0), (1, 0), (0, 8), (1, 8), (0, 4), (1, 4), (1, 2)]
    res = time_variants(name, lambda: ctx.prepare_count_matrix(batch, g[:, :32], batch, g[:, 32:], F, fidx), nbytes, variants)
    out["sets"][name] = {"shards": n, "encoded_bytes": int(nbytes), "kernel": "k_count_matrix_fusedq", "variants": res}
    batch.free()
    F.free()
if nd:
    rng = np.random.default_rng(11)
    rows = rng.integers(0, 2**64, (nd * 65, 16, 1024), dtype=np.uint64)
    batch = ctx.upload_dense(rows)
    del rows
    idx = np.arange(nd * 65).reshape(nd, 65)
    ga, gb, gf = idx[:, :32].copy(), idx[:, 32:64].copy(), idx[:, 64].copy()
    nbytes = nd * 65 * 16 * 8192
    variants = [(0, 0), (1, 0), (0, 2), (1, 2), (1, 8), (1, 16)]
    res = time_variants("dense", lambda: ctx.prepare_count_matrix(batch, ga, batch, gb, batch, gf), nbytes, variants)
    out["sets"]["dense"] = {"shards": nd, "encoded_bytes": int(nbytes), "kernel": "k_count_matrix_mfma", "variants": res}
    batch.free()
print(json.dumps(out))
