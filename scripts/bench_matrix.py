"""Dense count matrices of several tiles (GroupBy over fields of many rows): the i8 and the FP4 matrix
instruction (option matrix_fp4) on the same inputs; kernel time by HIP events around the call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from featurebase_amd.roaring import Context  # noqa: E402

ctx = Context(0)
st = torch.cuda.Stream()
ctx.set_stream(st.cuda_stream)


def rnd(n_rows, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randint(-(2**63), 2**63 - 1, (n_rows, 16, 1024), dtype=torch.int64, device="cuda", generator=g).cpu().numpy().view(np.uint64)


def t(fn, iters=8):
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
    return sorted(ts)[len(ts) // 2]


for n_shards, n_a, n_b in ((128, 32, 32), (64, 128, 128), (32, 256, 256), (64, 64, 64), (64, 96, 40)):
    wa, wb, wf = rnd(n_shards * n_a, 1), rnd(n_shards * n_b, 2), rnd(n_shards, 3)
    A, B, F = ctx.upload_dense(wa), ctx.upload_dense(wb), ctx.upload_dense(wf)
    ra, rb, rf = np.arange(n_shards * n_a).reshape(n_shards, n_a), np.arange(n_shards * n_b).reshape(n_shards, n_b), np.arange(n_shards)
    ref = None
    line = f"{n_shards:4d} shards x {n_a:3d} x {n_b:3d}:"
    for fp4 in (0, 1):
        ctx.set_option("matrix_fp4", fp4)
        tot = ctx.count_matrix(A, ra, B, rb, F, rf)
        if ref is None:
            ref = tot
            exp = int(sum(np.bitwise_count(wa[s * n_a + 1] & wb[s * n_b + 2] & wf[s]).sum() for s in range(n_shards)))
            assert int(tot[1, 2]) == exp, (int(tot[1, 2]), exp)
        ok = bool((tot == ref).all())
        us = t(lambda: ctx.count_matrix(A, ra, B, rb, F, rf))
        mfma_i8 = n_shards * 16 * ((n_a + 31) // 32) * ((n_b + 31) // 32) * 2048  # v_mfma_i32_32x32x32_i8 count
        line += f"  fp4={fp4} {us:8.1f} us ({'ok' if ok else 'MISMATCH'}; {mfma_i8 * 32 / 1024 / 2400:.0f} us of pure i8 MFMA issue)"
    print(line)
    ctx.set_option("matrix_fp4", -1)
    for b in (A, B, F):
        b.free()
