"""Structural fuzz of the C ABI: random batches (every encoding, missing slots, empty rows), random row index lists
(repeats, both operands from one batch or from two), random group shapes / matrix shapes / options, every entry point
of the hot path — compared with set algebra on the oracle containers' bit content (the per-container semantics are
pinned elsewhere: golden tables, test_gpu_parity.py; this file goes after indexing, scheduling and path-selection
bugs; tests/test_gpu_fuzz.py is the chained-operation fuzz).  FBK_FUZZ_ITERS=<n> runs more iterations, FBK_TEST_SEED
re-rolls them (scripts/fuzz_parity.sh)."""
import os

import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu
ITERS = int(os.environ.get("FBK_FUZZ_ITERS", "6"))
WIDTH = 1 << 20
popc = lambda w: int(np.bitwise_count(w).sum())  # noqa: E731


def make_batch(ctx, rng, n_rows):
    """-> (batch, words [n_rows, 16, 1024])"""
    p_missing = float(rng.choice([0.0, 0.15, 0.6, 0.95]))
    rows = [D.random_row(rng, 0, p_missing) if rng.random() > 0.05 else {} for _ in range(n_rows)]
    W = np.zeros((n_rows, 16, 1024), dtype=np.uint64)
    for r, row in enumerate(rows):
        for k, c in row.items():
            W[r, k & 15] = c.words()
    return ctx.upload([D.to_fbk_row(r) for r in rows]), W


def out_words(batch, n):
    res = batch.download()
    assert len(res) == n
    W = np.zeros((n, 16, 1024), dtype=np.uint64)
    for r, row in enumerate(res):
        for k, c in row.items():
            assert c.n == popc(c.words()) and c.n > 0, (r, k)  # empty results are nil slots
            W[r, k & 15] = c.words()
    return W, res


def check_optimized(O, res):
    """FBK_SETOP_OPTIMIZE: every container has the encoding Container.optimize() picks for its content."""
    for row in res:
        for c in row.values():
            oc = O.optimize(O.OContainer.bitmap(c.words()))
            assert c.n and c.typ == oc.typ and c.n == oc.n
            assert np.array_equal(np.asarray(c.data).reshape(-1), np.asarray(oc.data()).reshape(-1))


NP_OPS = {L.OP_AND: np.bitwise_and, L.OP_OR: np.bitwise_or, L.OP_XOR: np.bitwise_xor, L.OP_ANDNOT: lambda a, b: a & ~b}


def fold(op, W, ids):
    acc = W[ids[0]].copy()
    for i in ids[1:]:
        acc = NP_OPS[op](acc, W[i])
    return acc


def as_int(w):
    """[16, 1024] words -> python int (bit i = column i of the row)"""
    return int.from_bytes(w.tobytes(), "little")


@pytest.mark.parametrize("gen", [1, 2, 3])
@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_pairwise_and_folds(gpu_ctx, oracle, it, gen):
    try:
        gpu_ctx.set_option("pair_kernels", gen)  # both generations of the pair kernels see every case
    except Exception:
        if gen != 3:
            raise
        pytest.skip("k_icount3 exists in -DFBK_EXPERIMENTS builds only")
    try:
        _fuzz_pairwise_and_folds(gpu_ctx, oracle, it)
    finally:
        gpu_ctx.set_option("pair_kernels", 0)


def _fuzz_pairwise_and_folds(gpu_ctx, oracle, it):
    O = oracle
    rng = D.rng_for(7000, it)
    X, WX = make_batch(gpu_ctx, rng, int(rng.integers(1, 50)))
    Y, WY = (X, WX) if rng.random() < 0.3 else make_batch(gpu_ctx, rng, int(rng.integers(1, 50)))
    n = int(rng.integers(1, 120))
    ia, ib = rng.integers(0, len(WX), n), rng.integers(0, len(WY), n)
    for name, val in (("setop_direct_encode", int(rng.integers(0, 3))), ("pair_wpb", int(rng.choice([0, 1, 4]))), ("dense_spb", int(rng.choice([1, 2, 4, 8, 16])))):
        gpu_ctx.set_option(name, val)
    try:
        got = gpu_ctx.intersection_count(X, ia, Y, ib)
        assert got.tolist() == [popc(WX[a] & WY[b]) for a, b in zip(ia, ib)]
        assert X.count(ia).tolist() == [popc(WX[a]) for a in ia]
        for op in NP_OPS:
            flags = L.SETOP_OPTIMIZE if rng.random() < 0.5 else 0
            out, cnt = gpu_ctx.setop(op, X, ia, Y, ib, flags)
            W, res = out_words(out, n)
            exp = NP_OPS[op](WX[ia], WY[ib])
            assert np.array_equal(W, exp), (op, flags)
            assert cnt.tolist() == [popc(e) for e in exp]
            if flags:
                check_optimized(O, res)
            # a result is an ordinary batch: feed it back in
            if rng.random() < 0.5:
                again = gpu_ctx.intersection_count(out, np.arange(n), Y, ib)
                assert again.tolist() == [popc(e & WY[b]) for e, b in zip(exp, ib)]
            out.free()
        # plans: the same pairs, count / total / accumulate forms
        plan = gpu_ctx.plan(X, ia, Y, ib)
        plan.intersection_count_total()
        counts, total = plan.read(want_total=True)
        assert counts.tolist() == got.tolist() and int(total) == int(got.sum())
        plan.free()
        # n-way folds over random groups
        g, k = int(rng.integers(1, 40)), int(rng.integers(1, 9 if rng.random() < 0.8 else 80))
        groups = rng.integers(0, len(WX), (g, k))
        F, WF = make_batch(gpu_ctx, rng, g)
        for op in NP_OPS:
            flags = L.SETOP_OPTIMIZE if rng.random() < 0.5 else 0
            out, cnt = gpu_ctx.fold_n(op, X, groups, flags)
            W, res = out_words(out, g)
            exp = np.stack([fold(op, WX, ids) for ids in groups])
            assert np.array_equal(W, exp), (op, g, k)
            assert cnt.tolist() == [popc(e) for e in exp]
            if flags:
                check_optimized(O, res)
            out.free()
            assert gpu_ctx.fold_n_intersection_count(op, X, groups).tolist() == [popc(e) for e in exp]
            rf = rng.permutation(g)
            assert gpu_ctx.fold_n_intersection_count(op, X, groups, F, rf).tolist() == [popc(e & WF[f]) for e, f in zip(exp, rf)]
        out, cnt = gpu_ctx.union_n(X, groups)
        assert np.array_equal(out_words(out, g)[0], np.stack([fold(L.OP_OR, WX, ids) for ids in groups]))
        out.free()
        F.free()
    finally:
        for name, val in (("setop_direct_encode", 2), ("pair_wpb", 0), ("dense_spb", 16)):
            gpu_ctx.set_option(name, val)
        if Y is not X:
            Y.free()
        X.free()


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_count_matrix_and_topk(gpu_ctx, it):
    rng = D.rng_for(7100, it)
    n_shards = int(rng.integers(1, 7))
    n_a = int(rng.integers(1, 40)) if rng.random() < 0.8 else int(rng.integers(40, 80))
    n_b = int(rng.integers(1, 40))
    X, WX = make_batch(gpu_ctx, rng, int(rng.integers(1, 90)))
    dense = rng.random() < 0.25  # bitmap-only operands take the dense kernels
    if dense:
        X.free()
        WX = rng.integers(0, 1 << 63, (int(rng.integers(1, 60)), 16, 1024), dtype=np.uint64) & rng.integers(0, 1 << 63, (1, 16, 1024), dtype=np.uint64)
        X = gpu_ctx.upload_dense(WX.reshape(-1))
    ra, rb = rng.integers(0, len(WX), (n_shards, n_a)), rng.integers(0, len(WX), (n_shards, n_b))
    use_f = rng.random() < 0.7
    F, WF = make_batch(gpu_ctx, rng, n_shards) if use_f else (None, None)
    rf = rng.permutation(n_shards) if use_f else None
    opts = {"matrix_fused": int(rng.integers(-1, 2)), "matrix_fp4": int(rng.integers(-1, 2)), "matrix_spb": int(rng.choice([0, 1, 2, 4, 8, 16])),
            "matrix_pass_kb": int(rng.choice([1 << 20, 64, 4]))}
    for name, val in opts.items():
        gpu_ctx.set_option(name, val)
    try:
        tot, ps = gpu_ctx.count_matrix(X, ra, X, rb, F, rf, per_shard=True)
        exp = np.zeros((n_shards, n_a, n_b), dtype=np.uint64)
        for s in range(n_shards):
            a = WX[ra[s]].reshape(n_a, -1)
            if use_f:
                a = a & WF[rf[s]].reshape(1, -1)
            b = WX[rb[s]].reshape(n_b, -1)
            for j in range(n_b):
                exp[s, :, j] = np.bitwise_count(a & b[j]).sum(axis=1)
        assert np.array_equal(ps, exp), opts
        assert np.array_equal(tot, exp.sum(axis=0))
        assert np.array_equal(gpu_ctx.count_matrix(X, ra, X, rb, F, rf), exp.sum(axis=0))
        # TopK over the same rows: |row ∩ filter| summed over the shards, count descending, index ascending, zeros dropped
        gpu_ctx.set_option("topk_device_sort", int(rng.integers(-1, 2)))
        per_row = np.zeros(n_a, dtype=np.uint64)
        for s in range(n_shards):
            a = WX[ra[s]].reshape(n_a, -1)
            per_row += np.bitwise_count(a & WF[rf[s]].reshape(1, -1) if use_f else a).sum(axis=1).astype(np.uint64)
        order = sorted((i for i in range(n_a) if per_row[i]), key=lambda i: (-int(per_row[i]), i))
        for k in (0, 1, int(rng.integers(1, n_a + 1))):
            idx, cnt = gpu_ctx.topk(X, ra, k, F, rf)
            e = order[:k] if k else order
            assert idx.tolist() == e and cnt.tolist() == [int(per_row[i]) for i in e], (k, opts)
    finally:
        for name, val in (("matrix_fused", -1), ("matrix_fp4", -1), ("matrix_spb", 0), ("matrix_pass_kb", 1 << 20), ("topk_device_sort", -1)):
            gpu_ctx.set_option(name, val)
        if F is not None:
            F.free()
        X.free()


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_row_transforms(gpu_ctx, oracle, it):
    """CountRange, Flip, Shift on random rows and random index lists, as arithmetic on 2^20-bit integers."""
    O = oracle
    rng = D.rng_for(7200, it)
    X, WX = make_batch(gpu_ctx, rng, int(rng.integers(1, 40)))
    n = int(rng.integers(1, 60))
    rows = rng.integers(0, len(WX), n)
    ints = [as_int(WX[r]) for r in rows]
    try:
        for _ in range(4):
            a, b = sorted(int(v) for v in rng.integers(0, WIDTH + 1, 2))
            if rng.random() < 0.3:
                a, b = (a >> 16) << 16, (b >> 16) << 16
            gpu_ctx.set_option("count_range_reference_quirk", 0)  # (the bit count; the reference-identical default is tested in test_gpu_parity.py)
            got = gpu_ctx.count_range(X, rows, a, b)
            gpu_ctx.set_option("count_range_reference_quirk", 1)
            mask = ((1 << b) - 1) ^ ((1 << a) - 1)
            assert got.tolist() == [bin(v & mask).count("1") for v in ints], (a, b)
        for _ in range(3):
            a, b = sorted(int(v) for v in rng.integers(0, WIDTH, 2))
            flags = L.SETOP_OPTIMIZE if rng.random() < 0.5 else 0
            out, cnt = gpu_ctx.flip(X, rows, a, b, flags)
            W, res = out_words(out, n)
            mask = ((1 << (b + 1)) - 1) ^ ((1 << a) - 1)
            assert [as_int(w) for w in W] == [v ^ mask for v in ints], (a, b)
            assert cnt.tolist() == [bin(v ^ mask).count("1") for v in ints]
            if flags:
                check_optimized(O, res)
            out.free()
        carry = rng.integers(0, len(WX), n).astype(np.uint32)
        carry[rng.random(n) < 0.4] = gpu_ctx.NO_ROW
        for flags in (0, L.SETOP_OPTIMIZE):
            out, cnt = gpu_ctx.shift(X, rows, carry, flags)
            W, res = out_words(out, n)
            exp = [((v << 1) & ((1 << WIDTH) - 1)) | (0 if c == gpu_ctx.NO_ROW else as_int(WX[c]) >> (WIDTH - 1)) for v, c in zip(ints, carry)]
            assert [as_int(w) for w in W] == exp
            assert cnt.tolist() == [bin(e).count("1") for e in exp]
            if flags:
                check_optimized(O, res)
            out.free()
    finally:
        X.free()


@pytest.mark.parametrize("it", range(ITERS))
def test_fuzz_bsi(gpu_ctx, oracle, it):
    """BSI fragments of random depth (every remainder of the kernels' plane pipelines), shard count, value distribution
    (so that planes come out as arrays, runs, bitmaps or nothing) — or dense-layout batches — with random filters and
    predicates: Sum, Range (all operations), Between, Min, Max and the one-pass Range + Sum, every kernel form, against
    the oracle's fragment.sum / rangeOp / rangeBetween / min / max."""
    from oracle import pybsi as B

    B._lib()
    O = oracle
    rng = D.rng_for(7300, it)
    depth = int(rng.integers(1, 65))
    n_sh = int(rng.integers(1, 5))
    lim = (1 << min(depth, 63)) - 1
    frags, filts = [], []
    dense = rng.random() < 0.3
    for s in range(n_sh):
        ncol = int(rng.choice([0, 3, 200, 5000, 20000]))
        span = int(rng.choice([1 << 20, 1 << 16, 70000]))
        cols = rng.choice(span, size=min(ncol, span), replace=False)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            mag = [int(rng.integers(0, lim + 1)) if lim < (1 << 62) else int(rng.integers(0, 1 << 62)) * 2 + int(rng.integers(0, 2)) for _ in cols]
        elif kind == 1:  # few distinct small values: sparse upper planes, run-structured lower ones
            mag = [min(lim, int(v)) for v in rng.integers(0, 7, size=len(cols))]
        else:  # clustered around one value
            c0 = int(rng.integers(0, lim + 1)) if lim < (1 << 62) else int(rng.integers(0, 1 << 62))
            mag = [min(lim, max(0, c0 + int(d))) for d in rng.integers(-50, 50, size=len(cols))]
        sign = np.where(rng.random(len(cols)) < rng.choice([0.0, 0.4, 1.0]), -1, 1)
        vals = {int(c): int(m) * int(g) for c, m, g in zip(cols, mag, sign)}
        frags.append(B.bsi_fragment_from_values(vals, depth))
        fr = D.random_row(rng, 0, p_missing=float(rng.choice([0.0, 0.5])))
        if len(cols) and rng.random() < 0.7:  # make sure the filter meets the values somewhere
            fr[int(cols[0]) >> 16] = O.OContainer.array(sorted({int(c) & 0xFFFF for c in cols if (int(c) >> 16) == (int(cols[0]) >> 16)})[:4000])
        filts.append(O.OBitmap.from_containers(list(fr.items())))
    if dense:
        w = np.zeros((n_sh, depth + 2, 16, 1024), dtype=np.uint64)
        for s, fr in enumerate(frags):
            for r, bm in enumerate(fr.rows):
                if bm is not None:
                    for k, c in bm.items():
                        w[s, r, k & 15] = c.words()
        batch = gpu_ctx.upload_dense(w.reshape(-1))
        base = np.arange(n_sh, dtype=np.uint32) * (depth + 2)
    else:
        rows, base = [], []
        for fr in frags:
            base.append(len(rows))
            for r, bm in enumerate(fr.rows):
                rows.append({r * 16 + (k & 15): D.to_fbk(c) for k, c in bm.items() if c.n} if bm is not None else {})
        batch, base = gpu_ctx.upload(rows), np.array(base, dtype=np.uint32)
    F = gpu_ctx.upload([{k & 15: D.to_fbk(c) for k, c in f.items() if c.n} for f in filts])
    rf = np.arange(n_sh)
    try:
        for use_f in (False, True):
            fa = (F, rf) if use_f else (None, None)
            sums, cnts = gpu_ctx.bsi_sum(batch, base, depth, *fa)
            mn, mnc = gpu_ctx.bsi_min(batch, base, depth, *fa)
            mx, mxc = gpu_ctx.bsi_max(batch, base, depth, *fa)
            for s, fr in enumerate(frags):
                f = filts[s] if use_f else None
                assert (int(sums[s]), int(cnts[s])) == B.bsi_sum(fr, f, use_f), ("sum", s, use_f, depth)
                assert (int(mn[s]), int(mnc[s])) == B.bsi_min(fr, f, depth), ("min", s, use_f, depth)
                assert (int(mx[s]), int(mxc[s])) == B.bsi_max(fr, f, depth), ("max", s, use_f, depth)
        some = [0, 1, -1, lim, -lim, lim + 1 if lim < (1 << 62) else lim, int(rng.integers(-lim, lim + 1)) if lim < (1 << 62) else int(rng.integers(-(1 << 62), 1 << 62))]
        for name, op in B.OPS.items():
            for p in [some[int(j)] for j in rng.choice(len(some), size=3, replace=False)]:
                flags = L.SETOP_OPTIMIZE if rng.random() < 0.5 else 0
                out, cnt = gpu_ctx.bsi_range(batch, base, L.BSI_OPS[name], depth, p, flags)
                res = out.download()
                rs, rc = gpu_ctx.bsi_range_sum(batch, base, L.BSI_OPS[name], depth, p)
                fs, fc = gpu_ctx.bsi_range_sum(batch, base, L.BSI_OPS[name], depth, p, F, rf)
                for s, fr in enumerate(frags):
                    e = B.bsi_range(fr, op, depth, p)
                    got = {k & 15: c for k, c in res[s].items()}
                    exp = {k & 15: c for k, c in e.items() if c.n}
                    assert set(got) == set(exp) and all((got[k].words() == exp[k].words()).all() for k in exp), (name, p, s, depth)
                    assert int(cnt[s]) == e.count()
                    assert (int(rs[s]), int(rc[s])) == B.bsi_sum(fr, e, True), ("range_sum", name, p, s, depth)
                    assert (int(fs[s]), int(fc[s])) == B.bsi_sum(fr, e.intersect(filts[s]), True), ("range_sum+filter", name, p, s, depth)
                out.free()
        lo, hi = sorted(int(v) for v in (rng.integers(-lim, lim + 1, 2) if lim < (1 << 62) else rng.integers(-(1 << 62), 1 << 62, 2)))
        out, cnt = gpu_ctx.bsi_range_between(batch, base, depth, lo, hi)
        res = out.download()
        bs, bc = gpu_ctx.bsi_range_between_sum(batch, base, depth, lo, hi)
        fs, fc = gpu_ctx.bsi_range_between_sum(batch, base, depth, lo, hi, F, rf)
        for s, fr in enumerate(frags):
            e = B.bsi_range_between(fr, depth, lo, hi)
            assert (int(bs[s]), int(bc[s])) == B.bsi_sum(fr, e, True), ("between_sum", lo, hi, s, depth, dense)
            assert (int(fs[s]), int(fc[s])) == B.bsi_sum(fr, e.intersect(filts[s]), True), ("between_sum+filter", lo, hi, s, depth, dense)
        for s, fr in enumerate(frags):
            e = B.bsi_range_between(fr, depth, lo, hi)
            assert int(cnt[s]) == e.count(), ("between", lo, hi, s, depth)
            got = {k & 15: c for k, c in res[s].items()}
            assert all((got[k & 15].words() == c.words()).all() for k, c in e.items() if c.n)
        out.free()
    finally:
        batch.free()
        F.free()
