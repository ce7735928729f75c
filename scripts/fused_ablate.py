"""What the program-driven count-matrix kernel (fbk_matrix_fusedq.hip.h) spends its stage on, by switching parts of it OFF in the
experiments build (option matrix_fused_ablate; the counts are WRONG then): 1 no consumer arithmetic, 2 no array items, 8 no bitmap
rows, 16 the producers only keep the barriers.  Kernel time from the library's events, prepared query.

    python scripts/fused_ablate.py [config=4] [shards=1024]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import _experiments  # noqa: E402

_experiments.use()
import numpy as np  # noqa: E402

import datagen as D  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if cfg == 4 else 256)
if cfg == 4:
    rows, ga, gb, filt, _ = D.config4_flat(n, mp="fork")
    g = np.concatenate([ga, gb], axis=1)
else:
    rows, g, filt = D.config3_flat(n, mp="fork")
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)
batch = ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
F = ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
fidx = np.arange(n)
out = {"config": cfg, "shards": n, "encoded_bytes": int(rows.bytes + filt.bytes), "variants": []}
for prog in (2,):  # (2 = the shipped kernel; rounds' earlier kernels were removed)
    for ab, what in ((0, "everything"), (1, "no consumer arithmetic"), (2, "no array items"), (8, "no bitmap rows"), (10, "no array items, no bitmap rows"),
                     (16, "producers: barriers only"), (17, "barriers only (consumers and producers)"), (3, "no consumer arithmetic, no array items")):
        ctx.set_option("matrix_fused_ablate", ab)
        q = ctx.prepare_count_matrix(batch, g[:, :32], batch, g[:, 32:], F, fidx)
        q.run()
        ctx.set_option("time_kernels", 1)
        ts = []
        for _ in range(8):
            q.run()
            torch.cuda.synchronize()
            ts.append(ctx.get_option("last_kernel_ns") / 1e3)
        ctx.set_option("time_kernels", 0)
        ts.sort()
        out["variants"].append({"program": prog, "ablate": ab, "what": what, "kernel_us": ts[len(ts) // 2], "min": ts[0]})
        print(out["variants"][-1], file=sys.stderr, flush=True)
        q.free()
ctx.set_option("matrix_fused_ablate", 0)
print(json.dumps(out))
