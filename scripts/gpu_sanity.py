"""First-contact GPU check: correctness of the fbk kernels vs a numpy dense-bitset model
and a quick timing sweep of the dense |A∩B| kernel.  Not a test (tests/ has those);
run with:  gpurun -- python scripts/gpu_sanity.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from featurebase_amd import lib as L  # noqa: E402
from featurebase_amd.roaring import Container, Context  # noqa: E402


def popcount(a):
    return int(np.bitwise_count(a).sum())


def rand_container(rng, kind):
    if kind == "array":
        n = int(rng.integers(1, 4096))
        return Container.array(np.sort(rng.choice(65536, n, replace=False)).astype(np.uint16))
    if kind == "bitmap":
        return Container.bitmap(rng.integers(0, 2**64, 1024, dtype=np.uint64) & rng.integers(0, 2**64, 1024, dtype=np.uint64))
    if kind == "run":
        k = int(rng.integers(1, 200))
        pts = np.sort(rng.choice(65536, 2 * k, replace=False))
        return Container.run([(int(pts[2 * i]), int(pts[2 * i + 1]) - 1 if pts[2 * i + 1] - 1 >= pts[2 * i] else int(pts[2 * i])) for i in range(k)])
    if kind == "full":
        return Container.run([(0, 65535)])
    raise ValueError(kind)


def main():
    rng = np.random.default_rng(1)
    ctx = Context(0)
    # ---- dense correctness
    nrows = 64
    wa = rng.integers(0, 2**64, (nrows, 16, 1024), dtype=np.uint64)
    wb = rng.integers(0, 2**64, (nrows, 16, 1024), dtype=np.uint64)
    A, B = ctx.upload_dense(wa), ctx.upload_dense(wb)
    rows = np.arange(nrows)
    exp = np.array([popcount(wa[i] & wb[i]) for i in range(nrows)], dtype=np.uint64)
    got = ctx.intersection_count(A, rows, B, rows)
    assert (got == exp).all(), (got[:4], exp[:4])
    assert (A.count(rows) == np.array([popcount(wa[i]) for i in range(nrows)], dtype=np.uint64)).all()
    for op, f in [(L.OP_AND, np.bitwise_and), (L.OP_OR, np.bitwise_or), (L.OP_XOR, np.bitwise_xor), (L.OP_ANDNOT, lambda x, y: x & ~y)]:
        O, cnt = ctx.setop(op, A, rows, B, rows[::-1].copy())
        for i in (0, 5, nrows - 1):
            e = f(wa[i], wb[nrows - 1 - i])
            assert cnt[i] == popcount(e), (op, i)
        out = O.download()
        for i in (0, 5, nrows - 1):
            e = f(wa[i], wb[nrows - 1 - i])
            for s in range(16):
                assert (out[i][i * 16 + s].words() == e[s]).all(), (op, i, s)
        O.free()
    print("dense ok")
    # ---- mixed correctness
    kinds = ["array", "bitmap", "run", "full", None]
    rows_a, rows_b = [], []
    for r in range(40):
        ra, rb = {}, {}
        for s in range(16):
            ka, kb = kinds[rng.integers(0, 5)], kinds[rng.integers(0, 5)]
            if ka:
                ra[r * 16 + s] = rand_container(rng, ka)
            if kb:
                rb[r * 16 + s] = rand_container(rng, kb)
        rows_a.append(ra)
        rows_b.append(rb)
    MA, MB = ctx.upload(rows_a), ctx.upload(rows_b)
    rr = np.arange(40)
    Z = np.zeros(1024, dtype=np.uint64)
    wA = [[rows_a[r][r * 16 + s].words() if r * 16 + s in rows_a[r] else Z for s in range(16)] for r in range(40)]
    wB = [[rows_b[r][r * 16 + s].words() if r * 16 + s in rows_b[r] else Z for s in range(16)] for r in range(40)]
    exp = np.array([sum(popcount(wA[r][s] & wB[r][s]) for s in range(16)) for r in range(40)], dtype=np.uint64)
    got = ctx.intersection_count(MA, rr, MB, rr)
    assert (got == exp).all(), (got, exp)
    expc = np.array([sum(popcount(wA[r][s]) for s in range(16)) for r in range(40)], dtype=np.uint64)
    assert (MA.count(rr) == expc).all()
    for op, f in [(L.OP_AND, np.bitwise_and), (L.OP_OR, np.bitwise_or), (L.OP_XOR, np.bitwise_xor), (L.OP_ANDNOT, lambda x, y: x & ~y)]:
        O, cnt = ctx.setop(op, MA, rr, MB, rr)
        out = O.download()
        for r in range(40):
            tot = 0
            for s in range(16):
                e = f(wA[r][s], wB[r][s])
                tot += popcount(e)
                c = out[r].get(r * 16 + s)
                g = c.words() if c is not None else Z
                assert (g == e).all(), (op, r, s)
            assert cnt[r] == tot, (op, r)
        O.free()
    print("mixed ok")
    # download round trip
    back = MA.download()
    for r in range(40):
        assert set(back[r].keys()) == set(rows_a[r].keys())
        for k, c in rows_a[r].items():
            assert back[r][k].typ == c.typ and (back[r][k].words() == c.words()).all()
    print("roundtrip ok")
    for b in (A, B, MA, MB):
        b.free()

    # ---- timing sweep (needs torch only for events on the shared stream)
    import torch

    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    nsh = 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    # generate on the device, copy through host once (upload is not what we time)
    ta = torch.randint(-(2**63), 2**63 - 1, (nsh, 16, 1024), dtype=torch.int64, device="cuda", generator=g)
    tb = torch.randint(-(2**63), 2**63 - 1, (nsh, 16, 1024), dtype=torch.int64, device="cuda", generator=g)
    ha, hb = ta.cpu().numpy().view(np.uint64), tb.cpu().numpy().view(np.uint64)
    A, B = ctx.upload_dense(ha), ctx.upload_dense(hb)
    rows = np.arange(nsh)
    plan = ctx.plan(A, rows, B, rows)
    exp_total = popcount(ha & hb)
    for name, fn in [("icount", plan.intersection_count), ("and+count", lambda: plan.setop(L.OP_AND))]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        rd = 2 * nsh * 16 * 8192
        wr = nsh * 16 * 8192 if name != "icount" else 0
        print(f"{name}: {ms*1e3:.1f} us/step  read {rd/ms/1e6:.1f} GB/s  total {(rd+wr)/ms/1e6:.1f} GB/s  spb={os.environ.get('FBK_DENSE_SPB','16')}")
        plan.total()
        cnt, tot = plan.read(want_total=True)
        assert tot == exp_total and int(cnt.sum()) == exp_total, (tot, exp_total)
    print("timing ok")


if __name__ == "__main__":
    main()
