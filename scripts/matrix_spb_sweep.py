"""The dense count matrix at the per-rank size of BASELINE configs[3] (1024 shards x 32 x 32 rows + filter): kernel time
by slots per block (option matrix_spb: 16 = one block per shard, plain stores; fewer = more, shorter blocks and atomic adds)
in an interleaved A/B inside one process.

    python scripts/matrix_spb_sweep.py [shards=1024] [rounds=6]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_a = n_b = 32
dev = torch.device("cuda", 0)
ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)


def gen(n_rows, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randint(-(2**63), 2**63 - 1, (n_rows, 16, 1024), dtype=torch.int64, device=dev, generator=g)


ta, tb, tf = gen(ns * n_a, 1), gen(ns * n_b, 2), gen(ns, 3)
torch.cuda.synchronize()
A, B, F = ctx.upload_dense_device(ta.data_ptr(), ns * n_a), ctx.upload_dense_device(tb.data_ptr(), ns * n_b), ctx.upload_dense_device(tf.data_ptr(), ns)
del ta, tb, tf
torch.cuda.empty_cache()
ra, rb, rf = np.arange(ns * n_a).reshape(ns, n_a), np.arange(ns * n_b).reshape(ns, n_b), np.arange(ns)
q = ctx.prepare_count_matrix(A, ra, B, rb, F, rf, keep_per_shard=False)
ctx.set_option("time_kernels", 1)
variants = [0, 16, 8, 4, 2, 1]
ref = None
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        ctx.set_option("matrix_spb", v)
        q.run()
        ctx.synchronize()
        if r == 0:
            tot = q.read()
            if ref is None:
                ref = tot
            assert (tot == ref).all(), v
        else:
            times[v].append(ctx.get_option("last_kernel_ns") / 1e3)
ctx.set_option("matrix_spb", 0)
nbytes = ns * (n_a + n_b + 1) * 16 * 8192
out = {"shards": ns, "bytes": nbytes, "kernel_us": {}}
for v in variants:
    t = sorted(times[v])
    med = t[len(t) // 2]
    out["kernel_us"]["heuristic" if v == 0 else f"spb{v}"] = {"median": med, "min": t[0], "max": t[-1], "frac_of_8TBps": nbytes / (med * 1e-6) / 8e12}
print(json.dumps(out, indent=1))
