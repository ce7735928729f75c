"""k_count_matrix_fusedq of several LIBRARY builds (build_variants/<name>/libfbk.so, scripts/build_variant.sh) on one box, the builds
alternating: config 3's rows (GroupBy 32 x 32 + filter, 256 shards) and config 4 as SURVEY 8d writes it (log-uniform densities,
1024 shards).  The rows are generated once (`gen`) and kept in /dev/shm for the `run` processes — one process per library, because
FBK_LIB_PATH is read at import.  Prepared queries, kernel time from the library's events, the counts of every build compared with
the first build's (a checksum travels in the JSON).

    python scripts/fused_lib_ab.py gen [shards3=256] [shards4=1024]
    FBK_LIB_PATH=build_variants/x/libfbk.so python scripts/fused_lib_ab.py run   >> out.jsonl
    python scripts/fused_lib_ab.py rm
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

SHM = "/dev/shm/fbk_fused_lib_ab"
what = sys.argv[1] if len(sys.argv) > 1 else "run"
if what == "rm":
    import shutil

    shutil.rmtree(SHM, ignore_errors=True)
    sys.exit(0)
if what == "gen":
    import datagen as D

    n3 = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n4 = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    os.makedirs(SHM, exist_ok=True)
    meta = {}
    if n3:
        r3, g3, f3 = D.config3_flat(n3, mp="fork")
        np.save(f"{SHM}/c3_d.npy", r3.descs()), np.save(f"{SHM}/c3_p.npy", r3.payload()), np.save(f"{SHM}/c3_g.npy", g3)
        np.save(f"{SHM}/c3_fd.npy", f3.descs()), np.save(f"{SHM}/c3_fp.npy", f3.payload())
        meta["c3"] = {"n": n3, "n_rows": int(r3.n_rows), "bytes": int(r3.bytes + f3.bytes)}
    if n4:
        r4, ga4, gb4, f4, _ = D.config4_flat(n4, mp="fork")
        np.save(f"{SHM}/c4_d.npy", r4.descs()), np.save(f"{SHM}/c4_p.npy", r4.payload()), np.save(f"{SHM}/c4_g.npy", np.concatenate([ga4, gb4], axis=1))
        np.save(f"{SHM}/c4_fd.npy", f4.descs()), np.save(f"{SHM}/c4_fp.npy", f4.payload())
        meta["c4"] = {"n": n4, "n_rows": int(r4.n_rows), "bytes": int(r4.bytes + f4.bytes)}
    json.dump(meta, open(f"{SHM}/meta.json", "w"))
    sys.exit(0)

import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

meta = json.load(open(f"{SHM}/meta.json"))
ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
for kv in sys.argv[2:]:  # option=value ...
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
out = {"lib": os.environ.get("FBK_LIB_PATH", "product"), "options": sys.argv[2:]}
for key in ("c3", "c4"):
    if key not in meta:
        continue
    m = meta[key]
    d, p, g = np.load(f"{SHM}/{key}_d.npy"), np.load(f"{SHM}/{key}_p.npy"), np.load(f"{SHM}/{key}_g.npy")
    fd, fp = np.load(f"{SHM}/{key}_fd.npy"), np.load(f"{SHM}/{key}_fp.npy")
    batch = ctx.upload_flat(d, p, m["n_rows"])
    F = ctx.upload_flat(fd, fp, m["n"])
    q = ctx.prepare_count_matrix(batch, g[:, :32], batch, g[:, 32:], F, np.arange(m["n"]))
    for _ in range(25):  # (the clock settles ~15 ms after a pause: profiles/r06_first_launches.txt)
        q.run()
    got = q.read()
    ctx.set_option("time_kernels", 1)
    ts = []
    for _ in range(30):
        q.run()
        torch.cuda.synchronize()
        ts.append(ctx.get_option("last_kernel_ns") / 1e3)
    ctx.set_option("time_kernels", 0)
    ts.sort()
    out[key] = {"kernel_us": round(ts[len(ts) // 2], 1), "min": round(ts[0], 1), "p90": round(ts[int(len(ts) * 0.9)], 1),
                "frac": round(m["bytes"] / (ts[len(ts) // 2] * 1e-6) / 8e12, 4), "counts_sha": hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:12]}
    q.free()
    batch.free()
    F.free()
print(json.dumps(out), flush=True)
