"""GPU parity for the query-level kernels (n-way union, GroupBy/TopK count matrix, BSI
Sum/Range, optimize() re-encode) against the oracle, bit-exact, through the C ABI."""
import json
import os

import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "fragment_bsi_cases.json")))
ZERO = np.zeros(1024, dtype=np.uint64)


@pytest.fixture(scope="module")
def B(oracle):
    from oracle import pybsi

    pybsi._lib()
    return pybsi


def fbk_row_of_bitmap(bm, key_base=0):
    """oracle OBitmap (keys 0..15) -> {key: fbk Container}"""
    return {key_base + k: D.to_fbk(c) for k, c in bm.items() if c.n} if bm is not None else {}


def row_words(row):
    """{key: fbk Container} -> [16, 1024] uint64"""
    w = np.zeros((16, 1024), dtype=np.uint64)
    for k, c in row.items():
        w[k & 15] = c.words()
    return w


def bitmap_words(bm):
    w = np.zeros((16, 1024), dtype=np.uint64)
    if bm is not None:
        for k, c in bm.items():
            w[k & 15] = c.words()
    return w


def assert_optimized_like_oracle(O, got_row, exp_bm):
    """With FBK_SETOP_OPTIMIZE every output container has the encoding Container.optimize()
    (roaring.go:3412-3461) gives its content, byte for byte."""
    exp = {k & 15: c for k, c in exp_bm.items() if c.n}
    got = {k & 15: c for k, c in got_row.items()}
    assert set(got) == set(exp)
    for s, c in exp.items():
        # optimize() of the bit content (via a bitmap container, so runs are maximal)
        oc = O.optimize(O.OContainer.bitmap(c.words()))
        g = got[s]
        assert g.typ == oc.typ, (s, g.typ, oc)
        assert g.n == oc.n
        assert np.array_equal(np.asarray(g.data).reshape(-1), np.asarray(oc.data()).reshape(-1)), s


# ---- set-ops with optimize() -------------------------------------------------------------------
def test_setop_optimize_matches_oracle_encoding(gpu_ctx, oracle):
    O = oracle
    rng = D.rng_for(31)
    n = 24
    rows_a = [D.random_row(rng, r) for r in range(n)]
    rows_b = [D.random_row(rng, r) for r in range(n)]
    A, Bt = gpu_ctx.upload([D.to_fbk_row(r) for r in rows_a]), gpu_ctx.upload([D.to_fbk_row(r) for r in rows_b])
    idx = np.arange(n)
    for op, name in [(L.OP_AND, "intersect"), (L.OP_OR, "union"), (L.OP_XOR, "xor"), (L.OP_ANDNOT, "difference")]:
        out, cnt = gpu_ctx.setop(op, A, idx, Bt, idx, flags=L.SETOP_OPTIMIZE)
        res = out.download()
        for r in range(n):
            a = O.OBitmap.from_containers(list(rows_a[r].items()))
            b = O.OBitmap.from_containers(list(rows_b[r].items()))
            e = {"intersect": a.intersect, "xor": a.xor}[name](b) if name in ("intersect", "xor") else (a.union(b) if name == "union" else a.difference(b))
            assert_optimized_like_oracle(O, res[r], e)
            assert int(cnt[r]) == e.count()
        out.free()
    A.free()
    Bt.free()


# ---- n-way union (config 3 shape) ------------------------------------------------------------------
def make_union_groups(rng, n_groups, k):
    rows, groups = [], []
    for g in range(n_groups):
        ids = []
        for i in range(k):
            d = D.zipf_density(i)
            row = {}
            for s in range(16):
                if rng.random() < 0.1:
                    continue
                c = D.mixed_container_for_density(rng, d, rng.random() < 0.25)
                if c is not None and c.n:
                    row[g * 16 + s] = c
            ids.append(len(rows))
            rows.append(row)
        groups.append(ids)
    return rows, np.array(groups, dtype=np.uint32)


def test_union_n_vs_oracle(gpu_ctx, oracle):
    O = oracle
    rng = D.rng_for(41)
    rows, groups = make_union_groups(rng, 6, 12)
    # one group gets a full container to exercise the short-circuit (roaring.go:1465)
    rows[groups[2][5]][2 * 16 + 3] = O.OContainer.run([(0, 65535)])
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    filt_rows = [D.random_row(rng, g) for g in range(len(groups))]
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in filt_rows])
    for flags in (0, L.SETOP_OPTIMIZE):
        out, cnt = gpu_ctx.union_n(batch, groups, flags)
        res = out.download()
        for g, ids in enumerate(groups):
            bms = [O.OBitmap.from_containers(list(rows[i].items())) for i in ids]
            exp = bms[0].union(*bms[1:])  # Bitmap.Union n-way (roaring.go:1272)
            assert int(cnt[g]) == exp.count()
            assert (row_words(res[g]) == bitmap_words(exp)).all(), g
            if flags:
                assert_optimized_like_oracle(O, res[g], exp)
        out.free()
    # fused Union-of-k then IntersectionCount
    fidx = np.arange(len(groups))
    got = gpu_ctx.union_n_intersection_count(batch, groups, F, fidx)
    got_nf = gpu_ctx.union_n_intersection_count(batch, groups)
    for g, ids in enumerate(groups):
        bms = [O.OBitmap.from_containers(list(rows[i].items())) for i in ids]
        u = bms[0].union(*bms[1:])
        f = O.OBitmap.from_containers(list(filt_rows[g].items()))
        assert int(got[g]) == u.intersection_count(f), g
        assert int(got_nf[g]) == u.count()
    # k == 1 and single-row groups behave like a copy
    out, cnt = gpu_ctx.union_n(batch, groups[:, :1])
    assert cnt.tolist() == [sum(c.n for c in rows[ids[0]].values()) for ids in groups]
    out.free()
    batch.free()
    F.free()


# ---- GroupBy / TopK count matrix (config 4 shape) ---------------------------------------------------------
def test_count_matrix_vs_oracle(gpu_ctx, oracle, B):
    O = oracle
    rng = D.rng_for(51)
    n_shards, n_a, n_b = 11, 6, 5  # n_shards not a multiple of 8, n_a not a multiple of the tile
    a_rows, b_rows, f_rows = [], [], []
    for s in range(n_shards):
        a_rows.append([D.random_row(rng, 0, p_missing=0.3) for _ in range(n_a)])
        b_rows.append([D.random_row(rng, 0, p_missing=0.3) for _ in range(n_b)])
        f_rows.append(D.random_row(rng, 0, p_missing=0.2))
    A = gpu_ctx.upload([D.to_fbk_row(r) for s in a_rows for r in s])
    Bt = gpu_ctx.upload([D.to_fbk_row(r) for s in b_rows for r in s])
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in f_rows])
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)

    def obm(row):
        return O.OBitmap.from_containers(list(row.items()))

    for with_filter in (False, True):
        if with_filter:
            tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
        else:
            tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, per_shard=True)
        exp_tot = np.zeros((n_a, n_b), dtype=np.uint64)
        for s in range(n_shards):
            fa = B.Fragment([obm(r) for r in a_rows[s]])
            fb = B.Fragment([obm(r) for r in b_rows[s]])
            e = B.groupby_counts(fa, fb, obm(f_rows[s]) if with_filter else None)
            assert (ps[s] == e).all(), (s, with_filter)
            exp_tot += e
        assert (tot == exp_tot).all()
    # TopK shape (doTopK, executor.go:2705): every row of a fragment against one filter row
    tot, ps = gpu_ctx.count_matrix(A, ra, F, rf.reshape(-1, 1), per_shard=True)
    for s in range(n_shards):
        e = B.topk_row_counts(B.Fragment([obm(r) for r in a_rows[s]]), obm(f_rows[s]))
        assert ps[s, :, 0].tolist() == e.tolist()
    for b in (A, Bt, F):
        b.free()


# ---- BSI ---------------------------------------------------------------------------------------
def upload_bsi(gpu_ctx, frags):
    """frags: list (one per shard) of pybsi.Fragment -> (batch, base_rows)"""
    rows, base = [], []
    for fr in frags:
        base.append(len(rows))
        for r, bm in enumerate(fr.rows):
            rows.append(fbk_row_of_bitmap(bm, key_base=r * 16))
    return gpu_ctx.upload(rows), np.array(base, dtype=np.uint32)


def check_range(gpu_ctx, B, frags, depth, batch, base, op, pred, flags=0):
    out, cnt = gpu_ctx.bsi_range(batch, base, L.BSI_OPS[op], depth, pred, flags)
    res = out.download()
    for s, fr in enumerate(frags):
        e = B.bsi_range(fr, B.OPS[op], depth, pred)
        assert (row_words(res[s]) == bitmap_words(e)).all(), (op, pred, s)
        assert int(cnt[s]) == e.count()
    out.free()


@pytest.mark.parametrize("case", CASES["range_cases"], ids=lambda c: c["test"])
def test_bsi_range_reference_cases_on_gpu(gpu_ctx, B, case):
    """The reference's TestFragment_Range literals evaluated by the HIP plane-program kernel."""
    depth = max(v[1] for v in case["values"])
    fr = B.bsi_fragment_from_values({v[0]: v[2] for v in case["values"]}, depth)
    batch, base = upload_bsi(gpu_ctx, [fr])
    for q in case["queries"]:
        if q["kind"] == "rangeOp":
            out, cnt = gpu_ctx.bsi_range(batch, base, L.BSI_OPS[q["op"]], q["depth"], q["pred"])
        elif q["kind"] == "rangeBetween":
            out, cnt = gpu_ctx.bsi_range_between(batch, base, q["depth"], q["lo"], q["hi"])
        else:
            continue  # the *Unsigned kernels are not entry points of the boundary
        got = []
        for k, c in sorted(out.download()[0].items()):
            bits = np.unpackbits(c.words().view(np.uint8), bitorder="little")
            got.extend(((k & 15) << 16) + int(v) for v in np.nonzero(bits)[0])
        assert got == q["exp"], (case["test"], q)
        assert int(cnt[0]) == len(q["exp"])
        out.free()
    batch.free()


def test_bsi_diagonal_sweeps_on_gpu(gpu_ctx, B):
    """TestFragmentBSIUnsigned / Signed (fragment_internal_test.go:3768-4275) on the GPU."""
    k = 6
    fu = B.bsi_fragment_from_values({i: i for i in range(1 << k)}, k)
    mn, mx = 1 - (1 << k), (1 << k) - 1
    fs = B.bsi_fragment_from_values({v - mn: v for v in range(mn, mx + 1)}, k)
    frags = [fu, fs]
    batch, base = upload_bsi(gpu_ctx, frags)
    for p in list(range(-8, 12)) + list(range(55, 72)) + [-63, -64, -65, -126, 126, 127, 128]:
        for op in ("LT", "LTE", "GT", "GTE", "EQ", "NEQ"):
            check_range(gpu_ctx, B, frags, k, batch, base, op, p)
    for lo, hi in [(-3, 5), (0, 63), (10, 10), (-63, -1), (-20, 20), (5, 200), (-200, -5), (7, 6), (0, 0), (1, 64)]:
        out, cnt = gpu_ctx.bsi_range_between(batch, base, k, lo, hi)
        res = out.download()
        for s, fr in enumerate(frags):
            e = B.bsi_range_between(fr, k, lo, hi)
            assert (row_words(res[s]) == bitmap_words(e)).all(), (lo, hi, s)
        out.free()
    batch.free()


def test_bsi_random_multi_shard_sum_and_range(gpu_ctx, B, oracle):
    O = oracle
    rng = D.rng_for(61)
    depth = 64
    frags, filts, vals_all = [], [], []
    for s in range(5):
        ncol = [40000, 3000, 200, 1 << 16, 7][s]
        cols = rng.choice(1 << 20, size=ncol, replace=False)
        mag = rng.integers(0, 1 << 62, size=ncol) * (1 if s != 3 else 0) + rng.integers(0, 1000, size=ncol)
        sign = np.where(rng.random(ncol) < 0.4, -1, 1)
        vals = {int(c): int(m) * int(g) for c, m, g in zip(cols, mag, sign)}
        vals_all.append(vals)
        frags.append(B.bsi_fragment_from_values(vals, depth))
        fcols = [int(c) for c in cols[:: 2 + s]] + [5, 70000]
        filts.append(B.row_from_columns(fcols))
    batch, base = upload_bsi(gpu_ctx, frags)
    F = gpu_ctx.upload([fbk_row_of_bitmap(f) for f in filts])
    sums, counts = gpu_ctx.bsi_sum(batch, base, depth)
    for s, fr in enumerate(frags):
        assert (int(sums[s]), int(counts[s])) == B.bsi_sum(fr, None, False), s
    sums, counts = gpu_ctx.bsi_sum(batch, base, depth, F, np.arange(5))
    for s, fr in enumerate(frags):
        assert (int(sums[s]), int(counts[s])) == B.bsi_sum(fr, filts[s], True), s
    med = sorted(vals_all[0].values())[len(vals_all[0]) // 2]
    for op, p in [("GT", med), ("LTE", med), ("GT", 0), ("LT", 0), ("GTE", -1), ("EQ", med), ("NEQ", med), ("GT", (1 << 63) - 1), ("LT", -(1 << 63)), ("GTE", -(1 << 63))]:
        check_range(gpu_ctx, B, frags, depth, batch, base, op, p)
    check_range(gpu_ctx, B, frags, depth, batch, base, "GT", med, flags=L.SETOP_OPTIMIZE)
    # Range(>k) then Sum over the result (config 5's pipeline): filter = the range output batch
    out, cnt = gpu_ctx.bsi_range(batch, base, L.BSI_GT, depth, med)
    sums, counts = gpu_ctx.bsi_sum(batch, base, depth, out, np.arange(5))
    for s, fr in enumerate(frags):
        e = B.bsi_range(fr, B.GT, depth, med)
        assert (int(sums[s]), int(counts[s])) == B.bsi_sum(fr, e, True), s
        assert int(counts[s]) == int(cnt[s])
    out.free()
    batch.free()
    F.free()


def test_bsi_range_sum_one_pass_equals_range_then_sum(gpu_ctx, B, oracle):
    """fbk_bsi_range_sum (one pass over the planes) == fbk_bsi_sum over fbk_bsi_range == the oracle's fragment.sum over
    fragment.rangeOp, per shard, bit for bit: every operation, predicates around the reference's special forms and
    around stored values, with and without an extra filter row, mixed plane encodings, a shard without values."""
    rng = D.rng_for(63)
    depth = 64
    frags, filts, vals_all = [], [], []
    for s in range(6):
        ncol = [30000, 2500, 150, 1 << 15, 9, 0][s]
        cols = rng.choice(1 << 20, size=ncol, replace=False)
        mag = rng.integers(0, 1 << 62, size=ncol) * (1 if s != 3 else 0) + rng.integers(0, 1000, size=ncol)
        sign = np.where(rng.random(ncol) < 0.45, -1, 1)
        vals = {int(c): int(m) * int(g) for c, m, g in zip(cols, mag, sign)}
        vals_all.append(vals)
        frags.append(B.bsi_fragment_from_values(vals, depth))
        filts.append(B.row_from_columns([int(c) for c in cols[:: 2 + s]] + [5, 70000]))
    batch, base = upload_bsi(gpu_ctx, frags)
    F = gpu_ctx.upload([fbk_row_of_bitmap(f) for f in filts])
    rf = np.arange(len(frags))
    svals = sorted(vals_all[0].values())
    preds = [svals[len(svals) // 2], svals[len(svals) // 4], svals[0], svals[-1], 0, 1, -1, 2, -2, 500, -500, (1 << 63) - 1, -(1 << 63), 1 << 62, -(1 << 62)]
    for name, op in B.OPS.items():
        for p in preds:
            sums, cnts = gpu_ctx.bsi_range_sum(batch, base, L.BSI_OPS[name], depth, p)
            fs, fc = gpu_ctx.bsi_range_sum(batch, base, L.BSI_OPS[name], depth, p, F, rf)
            # ... and the two calls it replaces: fbk_bsi_range, then fbk_bsi_sum over the result rows
            rows2, _ = gpu_ctx.bsi_range(batch, base, L.BSI_OPS[name], depth, p)
            s2, c2 = gpu_ctx.bsi_sum(batch, base, depth, rows2, rf)
            rows2.free()
            for s, fr in enumerate(frags):
                e = B.bsi_range(fr, op, depth, p)
                assert (int(sums[s]), int(cnts[s])) == B.bsi_sum(fr, e, True), (name, p, s)
                assert (int(s2[s]), int(c2[s])) == (int(sums[s]), int(cnts[s])), (name, p, s, "range then sum")
                assert (int(fs[s]), int(fc[s])) == B.bsi_sum(fr, e.intersect(filts[s]), True), (name, p, s, "filter")
    # a shallower field, whose predicates saturate
    d2 = 10
    v2 = {int(c): int(v) for c, v in zip(rng.choice(1 << 20, size=4000, replace=False), rng.integers(-1023, 1024, size=4000))}
    fr2 = B.bsi_fragment_from_values(v2, d2)
    b2, base2 = upload_bsi(gpu_ctx, [fr2])
    for name, op in B.OPS.items():
        for p in (-1024, -1023, -1022, -512, -3, 3, 511, 512, 1022, 1023, 1024, 5000):
            sums, cnts = gpu_ctx.bsi_range_sum(b2, base2, L.BSI_OPS[name], d2, p)
            assert (int(sums[0]), int(cnts[0])) == B.bsi_sum(fr2, B.bsi_range(fr2, op, d2, p), True), (name, p)
    for b in (batch, F, b2):
        b.free()


def test_bsi_dense_batches_half_container_kernels(gpu_ctx, B, oracle):
    """A BSI batch in the dense layout (fbk_batch_upload_dense) takes the half-container-per-wavefront kernel for the
    one-pass Range + Sum: same totals as the oracle, with
    filters of every encoding, empty halves, a shard whose exists row is empty (Sum on the dense batch alongside)."""
    O = oracle
    rng = D.rng_for(64)
    n_sh, depth = 4, 13
    w = rng.integers(0, 1 << 63, (n_sh, depth + 2, 16, 1024), dtype=np.uint64) * 2 + rng.integers(0, 2, (n_sh, depth + 2, 16, 1024), dtype=np.uint64)
    w[:, 0] &= rng.integers(0, 1 << 63, (n_sh, 16, 1024), dtype=np.uint64)  # exists: about half of the columns
    w[:, 0, 3, 512:] = 0   # second half of a container without values
    w[:, 0, 4, :512] = 0   # first half
    w[:, 0, 5] = 0         # a whole container
    w[2, 0] = 0            # a shard without values
    w[:, 5] = 0            # an empty plane
    batch = gpu_ctx.upload_dense(w.reshape(-1))
    base = np.arange(n_sh, dtype=np.uint32) * (depth + 2)
    frags = [B.Fragment([O.OBitmap.from_containers([(sl, O.OContainer.bitmap(w[s, r, sl])) for sl in range(16) if w[s, r, sl].any()]) for r in range(depth + 2)]) for s in range(n_sh)]
    filt_rows = [D.random_row(rng, 0, p_missing=0.2) for _ in range(n_sh)]
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in filt_rows])
    fbms = [O.OBitmap.from_containers(list(r.items())) for r in filt_rows]
    rf = np.arange(n_sh)
    hw = "dense"
    sums, cnts = gpu_ctx.bsi_sum(batch, base, depth)
    fs, fc = gpu_ctx.bsi_sum(batch, base, depth, F, rf)
    for s in range(n_sh):
        assert (int(sums[s]), int(cnts[s])) == B.bsi_sum(frags[s], None, False), (s, hw)
        assert (int(fs[s]), int(fc[s])) == B.bsi_sum(frags[s], fbms[s], True), (s, hw)
    for name, op in B.OPS.items():
        for p in (100, -100, 4000, -4000, 8191, -8191, 1, -1, 0):
            sums, cnts = gpu_ctx.bsi_range_sum(batch, base, L.BSI_OPS[name], depth, p)
            fs, fc = gpu_ctx.bsi_range_sum(batch, base, L.BSI_OPS[name], depth, p, F, rf)
            for s in range(n_sh):
                e = B.bsi_range(frags[s], op, depth, p)
                assert (int(sums[s]), int(cnts[s])) == B.bsi_sum(frags[s], e, True), (name, p, s, hw)
                assert (int(fs[s]), int(fc[s])) == B.bsi_sum(frags[s], e.intersect(fbms[s]), True), (name, p, s, hw, "filter")
    # lo <= v <= hi: two lanes of one sign class (split at the highest differing bit), two sign classes, bounds beyond
    # the field; lo >= hi takes the two-pass path
    for lo, hi in ((100, 4000), (-4000, -100), (-300, 500), (0, 8191), (-8191, 0), (-9000, 9000), (1, 2), (4095, 4096), (-1, 0), (7, 7), (9, 3), (5000, 20000)):
        bs, bc = gpu_ctx.bsi_range_between_sum(batch, base, depth, lo, hi)
        fs, fc = gpu_ctx.bsi_range_between_sum(batch, base, depth, lo, hi, F, rf)
        for s in range(n_sh):
            e = B.bsi_range_between(frags[s], depth, lo, hi)
            assert (int(bs[s]), int(bc[s])) == B.bsi_sum(frags[s], e, True), ("between", lo, hi, s, hw)
            assert (int(fs[s]), int(fc[s])) == B.bsi_sum(frags[s], e.intersect(fbms[s]), True), ("between", lo, hi, s, hw, "filter")
    batch.free()
    F.free()


def test_bsi_minmax_reference_cases_on_gpu(gpu_ctx, B):
    """TestFragment_MinMax (fragment_internal_test.go:524-604) through fbk_bsi_min / fbk_bsi_max."""
    mc = CASES["minmax_case"]
    depth = mc["depth"]
    fr = B.bsi_fragment_from_values({v[0]: v[2] for v in mc["values"]}, depth)
    batch, base = upload_bsi(gpu_ctx, [fr])
    for kind, fn in (("min", gpu_ctx.bsi_min), ("max", gpu_ctx.bsi_max)):
        for t in mc[kind]:
            if t["filter"] is None:
                v, c = fn(batch, base, depth)
            else:
                F = gpu_ctx.upload([fbk_row_of_bitmap(B.row_from_columns(t["filter"]))])
                v, c = fn(batch, base, depth, F, [0])
                F.free()
            assert (int(v[0]), int(c[0])) == (t["exp"], t["cnt"]), (kind, t)
    batch.free()


def test_bsi_minmax_random_multi_shard_vs_oracle(gpu_ctx, B):
    """Mixed-encoding planes, several shards per launch, with and without a filter; includes
    all-negative / all-positive shards, a shard with no values, depth 64 and depth 1."""
    rng = D.rng_for(77)
    for depth in (1, 7, 20, 63, 64):
        frags, filts = [], []
        for s in range(6):
            ncol = [50000, 2000, 150, 1 << 16, 9, 0][s]
            cols = rng.choice(1 << 20, size=ncol, replace=False)
            hi = (1 << min(depth, 62))
            mag = rng.integers(0, hi, size=ncol)
            if s == 3:
                mag = mag % 7  # many ties, sparse high planes
            sign = {0: np.where(rng.random(ncol) < 0.5, -1, 1), 1: np.ones(ncol, dtype=np.int64), 2: -np.ones(ncol, dtype=np.int64)}.get(s % 3)
            vals = {int(c): int(m) * int(g) for c, m, g in zip(cols, mag, sign)}
            if depth == 64 and s == 1:
                vals[int(cols[0])] = (1 << 63) + 5  # magnitude bit 63 set: int64 wrap (fragment.go:795, 842)
            frags.append(B.bsi_fragment_from_values(vals, depth))
            filts.append(B.row_from_columns([int(c) for c in cols[:: 2 + s]] + [5, 70000]))
        batch, base = upload_bsi(gpu_ctx, frags)
        F = gpu_ctx.upload([fbk_row_of_bitmap(f) for f in filts])
        try:
            for fn, ofn in ((gpu_ctx.bsi_min, B.bsi_min), (gpu_ctx.bsi_max, B.bsi_max)):  # one wavefront per (shard, slot) + host fold
                v, c = fn(batch, base, depth)
                for s, fr in enumerate(frags):
                    assert (int(v[s]), int(c[s])) == ofn(fr, None, depth), (depth, s)
                v, c = fn(batch, base, depth, F, np.arange(len(frags)))
                for s, fr in enumerate(frags):
                    assert (int(v[s]), int(c[s])) == ofn(fr, filts[s], depth), (depth, s, "filtered")
        finally:
            pass
        batch.free()
        F.free()


def test_fold_n_vs_oracle(gpu_ctx, oracle):
    """n-way Intersect / Xor / Difference (and Union again) in one launch against the oracle's
    left fold of Bitmap.Intersect / Xor / Difference(others...) — what executeIntersectShard,
    executeXorShard and executeDifferenceShard compute child by child."""
    O = oracle
    rng = D.rng_for(43)
    rows, groups = make_union_groups(rng, 6, 7)
    # full containers (identity of AND, saturate OR / the subtrahend of ANDNOT) and nil slots
    rows[groups[1][0]][1 * 16 + 3] = O.OContainer.run([(0, 65535)])
    rows[groups[1][2]][1 * 16 + 3] = O.OContainer.run([(0, 65535)])
    rows[groups[2][4]][2 * 16 + 5] = O.OContainer.run([(0, 65535)])
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    filt_rows = [D.random_row(rng, g) for g in range(len(groups))]
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in filt_rows])

    def fold(op, bms):
        if op == L.OP_OR:
            return bms[0].union(*bms[1:])
        if op == L.OP_ANDNOT:
            return bms[0].difference(*bms[1:]) if len(bms) > 1 else bms[0]
        acc = bms[0]
        for b in bms[1:]:
            acc = acc.intersect(b) if op == L.OP_AND else acc.xor(b)
        return acc

    for kk in (7, 2, 1):
        gs = groups[:, :kk]
        for op in (L.OP_AND, L.OP_OR, L.OP_XOR, L.OP_ANDNOT):
            out, cnt = gpu_ctx.fold_n(op, batch, gs, L.SETOP_OPTIMIZE if kk == 7 else 0)
            res = out.download()
            got = gpu_ctx.fold_n_intersection_count(op, batch, gs, F, np.arange(len(gs)))
            got_nf = gpu_ctx.fold_n_intersection_count(op, batch, gs)
            for g, ids in enumerate(gs):
                bms = [O.OBitmap.from_containers(list(rows[i].items())) for i in ids]
                exp = fold(op, bms)
                assert int(cnt[g]) == exp.count(), (op, kk, g)
                assert (row_words(res[g]) == bitmap_words(exp)).all(), (op, kk, g)
                if kk == 7:
                    assert_optimized_like_oracle(O, res[g], exp)
                f = O.OBitmap.from_containers(list(filt_rows[g].items()))
                assert int(got[g]) == exp.intersection_count(f), (op, kk, g)
                assert int(got_nf[g]) == exp.count()
            out.free()
    with pytest.raises(L.FbkError):
        gpu_ctx.fold_n(L.OP_AND, batch, np.zeros((2, 0), dtype=np.uint32))
    batch.free()
    F.free()


def test_count_range_vs_oracle(gpu_ctx, oracle):
    """Bitmap.CountRange (roaring.go:573) per row: container-aligned ranges (what fragment.go
    uses), ranges inside one container and ranges straddling several, on mixed encodings."""
    O = oracle
    rng = D.rng_for(47)
    rows = [D.random_row(rng, r) for r in range(12)]
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    idx = np.arange(len(rows))
    words = [row_words(D.to_fbk_row(r)) for r in rows]
    ranges = [(0, 1 << 20), (0, 0), (65536, 131072), (5, 6), (100, 65536 + 77), (65535, 65537), (3 * 65536 + 12345, 9 * 65536 + 1),
              (1 << 19, 1 << 20), ((1 << 20) - 1, 1 << 20), (64, 128), (63, 129), (1000, 1000)]
    ranges += [tuple(sorted(int(x) for x in rng.integers(0, (1 << 20) + 1, size=2))) for _ in range(20)]
    obms = [O.OBitmap.from_containers(sorted((k & 15, c) for k, c in r.items())) for r in rows]
    for s, e in ranges:
        # the default is the reference's number (RunCountRange's double count included, roaring.go:3216-3227) ...
        got = gpu_ctx.count_range(batch, idx, s, e)
        assert got.tolist() == [bm.count_range(s, e) for bm in obms], (s, e)
        # ... option count_range_reference_quirk = 0 the bits of [s, e)
        gpu_ctx.set_option("count_range_reference_quirk", 0)
        try:
            got = gpu_ctx.count_range(batch, idx, s, e)
        finally:
            gpu_ctx.set_option("count_range_reference_quirk", 1)
        for r in range(len(rows)):
            bits = np.unpackbits(words[r].reshape(-1).view(np.uint8), bitorder="little")
            truth = int(bits[s:e].sum())
            assert int(got[r]) == truth, (r, s, e)
            # the reference's own arithmetic, wherever RunCountRange's Last == end quirk
            # (roaring.go:3216-3227) cannot fire: no run container ends exactly at `end`
            bm = O.OBitmap.from_containers([(k & 15, c) for k, c in rows[r].items()])
            quirk = any(c.typ == 3 and any(int(l) == (e & 0xFFFF) for _, l in c.data().reshape(-1, 2)) for _, c in bm.items())
            if not quirk:
                assert bm.count_range(s, e) == truth, (r, s, e)
    with pytest.raises(L.FbkError):
        gpu_ctx.count_range(batch, idx, 5, 4)
    batch.free()


@pytest.mark.parametrize("n_shards,n_a,n_b,use_filter", [(3, 32, 32, True), (2, 37, 45, False), (5, 5, 3, True), (1, 1, 2, False), (9, 64, 33, True), (2, 40, 8, True), (2, 8, 70, False), (1, 100, 97, True)])
def test_count_matrix_dense_kernel_vs_numpy(gpu_ctx, n_shards, n_a, n_b, use_filter):
    """The all-bitmap fast path of fbk_count_matrix (k_count_matrix_mfma: bits expanded to i8
    bytes, v_mfma_i32_32x32x32_i8 tiles, per-wave DMA ring) against numpy popcounts: full
    matrix, per shard and in total, ragged tile edges, with and without the filter row; every
    register tiling of the kernel (1 x 1, 2 x 1, 1 x 2, 2 x 2 tiles of 32 rows per block)."""
    wa = D.dense_rows(n_shards * n_a, 0.3, 811)
    wb = D.dense_rows(n_shards * n_b, 0.6, 812)
    wf = D.dense_rows(n_shards, 0.5, 813)
    wa[-1] = 0  # an empty row
    wb[0] = np.uint64(0xFFFFFFFFFFFFFFFF)  # a full row
    A, Bt = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb)
    F = gpu_ctx.upload_dense(wf) if use_filter else None
    rng = np.random.default_rng(5)
    ra = np.stack([rng.permutation(n_a) + s * n_a for s in range(n_shards)])  # rows in any order
    rb = np.stack([rng.permutation(n_b) + s * n_b for s in range(n_shards)])
    rf = np.arange(n_shards)
    tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf if use_filter else None, per_shard=True)
    exp = np.zeros((n_shards, n_a, n_b), dtype=np.uint64)
    for s in range(n_shards):
        for i in range(n_a):
            x = wa[ra[s, i]] & wf[s] if use_filter else wa[ra[s, i]]
            for j in range(n_b):
                exp[s, i, j] = np.bitwise_count(x & wb[rb[s, j]]).sum()
    assert (ps == exp).all()
    assert (tot == exp.sum(axis=0)).all()
    # every slots-per-block split of the launch (16 = whole shards with plain stores, smaller =
    # atomic adds of partial matrices), and the vector-ALU kernel kept for A/B measurements
    # ... on both matrix instructions: v_mfma_i32_32x32x32_i8 (bit -> byte) and the block-scaled FP4
    # v_mfma_scale_f32_32x32x64_f8f6f4 (bit -> nibble, option matrix_fp4; the default picks it for
    # matrices of several tiles)
    try:
        for fp4 in (0, 1):
            gpu_ctx.set_option("matrix_fp4", fp4)
            for spb in ("16", "8", "4", "2", "1"):
                gpu_ctx.set_option("matrix_spb", int(spb))
                tot2, ps2 = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf if use_filter else None, per_shard=True)
                assert (ps2 == exp).all(), (fp4, spb)
                assert (tot2 == tot).all(), (fp4, spb)
    finally:
        gpu_ctx.set_option("matrix_spb", 0)
        gpu_ctx.set_option("matrix_fp4", -1)
    # the total alone (no per-shard matrices asked for) is reduced in passes over the shards: force
    # passes of one or two shards
    try:
        gpu_ctx.set_option("matrix_pass_kb", int(str(max(1, (2 * n_a * n_b * 8) // 1024))))
        tot3 = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf if use_filter else None)
        assert (tot3 == tot).all()
    finally:
        gpu_ctx.set_option("matrix_pass_kb", 1048576)
    A.free()
    Bt.free()
    if F is not None:
        F.free()


def test_count_matrix_dense_by_ticket_vs_numpy(gpu_ctx):
    """Round 6: a single-tile dense count matrix of many shards takes its units by TICKET (option matrix_tickets, default 1:
    three tiers of unit sizes, spare blocks that find no unit — fbk_matrix_mfma.hip.h).  520 shards (not a multiple of anything)
    x 3 x 2 rows + filter against numpy popcounts, with the launch by block id (matrix_tickets = 0) beside it, for every
    slots-per-block value: 0 (the library's choice: 2 here), 2 and 4 have a ticket plan at this size, 8 and 16 do not — and
    twice in a row, since the counter has to be back at zero after every launch."""
    n_shards, n_a, n_b = 520, 3, 2
    wa = D.dense_rows(n_shards * n_a, 0.5, 911)
    wb = D.dense_rows(n_shards * n_b, 0.5, 912)
    wf = D.dense_rows(n_shards, 0.5, 913)
    wa[7] = 0  # an empty row, a full one
    wb[-1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    A, Bt, F = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb), gpu_ctx.upload_dense(wf)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)
    exp = np.zeros((n_shards, n_a, n_b), dtype=np.uint64)
    for i in range(n_a):
        x = wa[i::n_a] & wf
        for j in range(n_b):
            exp[:, i, j] = np.bitwise_count(x & wb[j::n_b]).sum(axis=(1, 2))
    try:
        for tickets in (1, 0):
            gpu_ctx.set_option("matrix_tickets", tickets)
            for spb in (0, 2, 4, 8, 16):
                gpu_ctx.set_option("matrix_spb", spb)
                for rep in range(2):
                    tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
                    assert (ps == exp).all(), (tickets, spb, rep)
                    assert (tot == exp.sum(axis=0)).all(), (tickets, spb, rep)
        # without the filter row, and the total alone in passes of ~130 shards
        gpu_ctx.set_option("matrix_tickets", 1)
        gpu_ctx.set_option("matrix_spb", 0)
        exp2 = np.zeros((n_a, n_b), dtype=np.uint64)
        for i in range(n_a):
            for j in range(n_b):
                exp2[i, j] = np.bitwise_count(wa[i::n_a] & wb[j::n_b]).sum()
        assert (gpu_ctx.count_matrix(A, ra, Bt, rb, None, None) == exp2).all()
        gpu_ctx.set_option("matrix_pass_kb", max(1, (130 * n_a * n_b * 8) // 1024))
        assert (gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf) == exp.sum(axis=0)).all()
    finally:
        gpu_ctx.set_option("matrix_tickets", 1)
        gpu_ctx.set_option("matrix_spb", 0)
        gpu_ctx.set_option("matrix_pass_kb", 1048576)
    A.free()
    Bt.free()
    F.free()


@pytest.mark.parametrize("n_shards", [256, 257, 383, 512, 513, 777, 1025])
def test_count_matrix_dense_ticket_tiers_at_their_edges(gpu_ctx, n_shards):
    """The tiers of the ticketed launch (mm_ticket_plan) at shard counts around the places where the plan changes: the first size with
    a plan at 2 slots per unit (256), odd sizes, the first with a plan at 4 slots (512), one past a power of two.  1 x 2 rows + filter,
    per-shard counts against numpy popcounts, units by ticket and by block id, library-chosen and forced slots per unit."""
    n_a, n_b = 1, 2
    wa = D.dense_rows(n_shards * n_a, 0.5, 3100 + n_shards)
    wb = D.dense_rows(n_shards * n_b, 0.5, 3200 + n_shards)
    wf = D.dense_rows(n_shards, 0.5, 3300 + n_shards)
    A, Bt, F = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb), gpu_ctx.upload_dense(wf)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)
    exp = np.zeros((n_shards, n_a, n_b), dtype=np.uint64)
    x = wa & wf
    for j in range(n_b):
        exp[:, 0, j] = np.bitwise_count(x & wb[j::n_b]).sum(axis=(1, 2))
    try:
        for tickets, spb in ((1, 0), (1, 2), (1, 4), (1, 8), (0, 0)):
            gpu_ctx.set_option("matrix_tickets", tickets)
            gpu_ctx.set_option("matrix_spb", spb)
            for rep in range(2):
                tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
                assert (ps == exp).all(), (n_shards, tickets, spb, rep)
                assert (tot == exp.sum(axis=0)).all(), (n_shards, tickets, spb, rep)
    finally:
        gpu_ctx.set_option("matrix_tickets", 1)
        gpu_ctx.set_option("matrix_spb", 0)
    A.free()
    Bt.free()
    F.free()


def test_rows_vs_filter_kernel_vs_oracle(gpu_ctx, oracle):
    """fbk_count_matrix with one B row per shard and no extra filter = doTopK / fragment.top:
    every encoding of the rows against every encoding of the filter, incl. full / empty / nil
    containers, arrays past 4096 values and long run lists; 150 rows so that several row chunks
    and a ragged last chunk are exercised."""
    O = oracle
    rng = D.rng_for(53)
    n_shards, n_a = 3, 150
    rows, ra = [], []
    for s in range(n_shards):
        ids = []
        for i in range(n_a):
            row = D.random_row(rng, s)
            if i % 17 == 0:
                row[s * 16 + 5] = O.OContainer.run([(0, 65535)])
            if i % 19 == 0:
                row[s * 16 + 6] = O.OContainer.array(np.sort(rng.choice(65536, size=6000, replace=False)))
            if i % 23 == 0:
                row[s * 16 + 7] = O.OContainer.run([(j * 20, j * 20 + 7) for j in range(3000)])
            ids.append(len(rows))
            rows.append(row)
        ra.append(ids)
    filt = []
    for s in range(n_shards):
        f = D.random_row(rng, s)
        f[s * 16 + 0] = O.OContainer.bitmap(rng.integers(0, 1 << 63, size=1024, dtype=np.uint64))
        f[s * 16 + 1] = O.OContainer.array(np.sort(rng.choice(65536, size=900, replace=False)))
        f[s * 16 + 2] = O.OContainer.run([(100, 40000), (50000, 50001)])
        f[s * 16 + 3] = O.OContainer.run([(0, 65535)])
        f.pop(s * 16 + 4, None)  # nil filter slot: nothing counts there
        filt.append(f)
    A = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in filt])
    tot, ps = gpu_ctx.count_matrix(A, np.array(ra), F, np.arange(n_shards).reshape(-1, 1), per_shard=True)
    exp = np.zeros((n_shards, n_a), dtype=np.uint64)
    for s in range(n_shards):
        for i in range(n_a):
            r = rows[ra[s][i]]
            exp[s, i] = sum(O.intersection_count(r[k], filt[s][k]) for k in r if k in filt[s])
    assert (ps[:, :, 0] == exp).all()
    assert (tot[:, 0] == exp.sum(axis=0)).all()
    A.free()
    F.free()


def test_fold_n_more_than_64_rows_per_group(gpu_ctx, oracle):
    """Groups of 150 rows: the descriptor window of the fold kernels (64 rows per pass) wraps
    twice and ends ragged; all four ops, with a full container in the last window."""
    O = oracle
    rng = D.rng_for(59)
    rows, groups = make_union_groups(rng, 2, 150)
    rows[groups[1][140]][1 * 16 + 9] = O.OContainer.run([(0, 65535)])
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    for op in (L.OP_AND, L.OP_OR, L.OP_XOR, L.OP_ANDNOT):
        out, cnt = gpu_ctx.fold_n(op, batch, groups)
        res = out.download()
        for g, ids in enumerate(groups):
            bms = [O.OBitmap.from_containers(list(rows[i].items())) for i in ids]
            if op == L.OP_OR:
                exp = bms[0].union(*bms[1:])
            elif op == L.OP_ANDNOT:
                exp = bms[0].difference(*bms[1:])
            else:
                exp = bms[0]
                for b in bms[1:]:
                    exp = exp.intersect(b) if op == L.OP_AND else exp.xor(b)
            assert int(cnt[g]) == exp.count(), (op, g)
            assert (row_words(res[g]) == bitmap_words(exp)).all(), (op, g)
        out.free()
    batch.free()


@pytest.fixture(params=[(1, 2048, 0), (1, 64, 0), (0, 2048, 0), (1, 2048, 20)],
                ids=["heavy-shadows", "shadows-of-arrays-over-64-values", "no-shadows", "shadows-of-runs-over-20-intervals"])
def shadow_mode(request, gpu_ctx):
    """Count matrix over encoded rows: run containers and long arrays as dense shadows built per batch on first use (default), the
    same with nearly every array shadowed (more than 42 bitmap rows per slot: the in-place path of the bitmap waves), every
    container decoded in every query (run rows, long arrays), and (round 6, option matrix_shadow_run) only the run containers of
    more than 20 intervals shadowed — shadowed and in-place run rows side by side in one slot.  Each result is compared with the
    oracle's groupByIterator counts per shard, and with the generic pair kernel on the same call (two kernels that share no code)."""
    gpu_ctx.set_option("matrix_shadow", request.param[0])
    gpu_ctx.set_option("matrix_shadow_array", request.param[1])
    gpu_ctx.set_option("matrix_shadow_run", request.param[2])
    yield request.param
    gpu_ctx.set_option("matrix_shadow", 1)
    gpu_ctx.set_option("matrix_shadow_array", 2048)
    gpu_ctx.set_option("matrix_shadow_run", 0)


def test_count_matrix_fused_duplicated_and_broadcast_rows(gpu_ctx, oracle, B):
    """Row lists may name a batch row any number of times (check_rows only range-checks): one row of long arrays listed 32 times
    on each side, and ONE filter row and one B row set broadcast to every shard.  The program of the matrix-core kernel emits an
    item per row OCCURRENCE (k_fused_program), far more than the payload-based first guess of its item area allows for: the
    build counts what it needs and launches again (fused_program_build).  Checked against the oracle and the generic kernel."""
    O = oracle
    rng = D.rng_for(58)
    n_shards, n = 6, 32

    def obm(row):
        return O.OBitmap.from_containers(list(row.items()))

    def long_arrays(seed):  # 16 arrays of ~2000 values: 16 items each per occurrence and stage range
        r = np.random.default_rng(seed)
        return {k: O.OContainer.array(np.sort(r.choice(65536, size=int(r.integers(1800, 2200)), replace=False)).astype(np.uint16)) for k in range(16)}

    a_rows = [long_arrays(100 + i) for i in range(3)]
    b_rows = [long_arrays(200 + i) for i in range(2)]
    f_row = D.random_row(rng, 0, p_missing=0.1)
    A, Bt, F = gpu_ctx.upload([D.to_fbk_row(r) for r in a_rows]), gpu_ctx.upload([D.to_fbk_row(r) for r in b_rows]), gpu_ctx.upload([D.to_fbk_row(f_row)])
    ra = np.zeros((n_shards, n), dtype=np.uint32)
    ra[:, ::5] = 1
    ra[3] = 2                      # every A row of shard 3 is batch row 2
    rb = np.ones((n_shards, n), dtype=np.uint32)
    rb[:, 7] = 0
    rf = np.zeros(n_shards, dtype=np.uint32)  # the one filter row, for every shard
    try:
        gpu_ctx.set_option("matrix_fused", 1)
        tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
        gpu_ctx.set_option("matrix_fused", 0)
        tot_g, ps_g = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
    finally:
        gpu_ctx.set_option("matrix_fused", -1)
    for s in range(n_shards):
        e = B.groupby_counts(B.Fragment([obm(a_rows[i]) for i in ra[s]]), B.Fragment([obm(b_rows[j]) for j in rb[s]]), obm(f_row))
        assert (ps[s] == e).all(), s
    assert (tot == ps.sum(axis=0)).all() and (tot_g == tot).all() and (ps_g == ps).all()
    for b in (A, Bt, F):
        b.free()


@pytest.mark.parametrize("a_dense,b_dense,f_mode", [(False, False, "mixed"), (True, False, "none"), (False, True, "dense"), (False, False, "none")])
def test_count_matrix_mixed_rows_fused_and_generic_paths(gpu_ctx, oracle, B, a_dense, b_dense, f_mode, shadow_mode):
    """nA x nB >= 16 with array / run containers among the rows: fbk_count_matrix decodes the rows inside the
    matrix-core kernel; every mix of dense and encoded operands, checked against the oracle's groupByIterator
    counts — and against the generic pair kernel (matrix_fused=0) on the same call."""
    O = oracle
    rng = D.rng_for(57)
    n_shards, n_a, n_b = 5, 48, 43

    def obm(row):
        return O.OBitmap.from_containers(list(row.items()))

    def side(n, dense, seed):
        if dense:
            w = D.dense_rows(n_shards * n, 0.3, seed).reshape(n_shards * n, 16, 1024)
            rows = [[{k: O.OContainer.bitmap(w[s * n + i, k]) for k in range(16)} for i in range(n)] for s in range(n_shards)]
            return gpu_ctx.upload_dense(w.reshape(-1)), rows
        rows = [[D.random_row(rng, 0, p_missing=0.25) for _ in range(n)] for s in range(n_shards)]
        return gpu_ctx.upload([D.to_fbk_row(r) for s in rows for r in s]), rows

    A, a_rows = side(n_a, a_dense, 571)
    Bt, b_rows = side(n_b, b_dense, 572)
    F, f_rows = None, None
    if f_mode == "mixed":
        f_rows = [D.random_row(rng, 0, p_missing=0.2) for _ in range(n_shards)]
        F = gpu_ctx.upload([D.to_fbk_row(r) for r in f_rows])
    elif f_mode == "dense":
        wf = D.dense_rows(n_shards, 0.6, 573).reshape(n_shards, 16, 1024)
        f_rows = [{k: O.OContainer.bitmap(wf[s, k]) for k in range(16)} for s in range(n_shards)]
        F = gpu_ctx.upload_dense(wf.reshape(-1))
    perm = np.random.default_rng(3).permutation(n_shards)  # shards in any order
    ra = np.stack([np.arange(n_a)[::-1] + s * n_a for s in perm])
    rb = np.stack([np.arange(n_b) + s * n_b for s in perm])
    tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, perm if F is not None else None, per_shard=True)
    exp_tot = np.zeros((n_a, n_b), dtype=np.uint64)
    for k, s in enumerate(perm):
        fa = B.Fragment([obm(r) for r in a_rows[s][::-1]])
        fb = B.Fragment([obm(r) for r in b_rows[s]])
        e = B.groupby_counts(fa, fb, obm(f_rows[s]) if F is not None else None)
        assert (ps[k] == e).all(), (k, s)
        exp_tot += e
    assert (tot == exp_tot).all()
    # the default above decodes inside the matrix-core kernel (fbk_matrix_fusedq.hip.h): every
    # slots-per-block split of that launch; then the same call on the generic pair kernel (matrix_fused=0)
    try:
        for spb in (16, 8, 4, 2, 1):
            gpu_ctx.set_option("matrix_spb", spb)
            tot_s, ps_s = gpu_ctx.count_matrix(A, ra, Bt, rb, F, perm if F is not None else None, per_shard=True)
            assert (tot_s == tot).all() and (ps_s == ps).all(), spb
        gpu_ctx.set_option("matrix_spb", 0)
        gpu_ctx.set_option("matrix_fused", 0)
        tot_g, ps_g = gpu_ctx.count_matrix(A, ra, Bt, rb, F, perm if F is not None else None, per_shard=True)
    finally:
        gpu_ctx.set_option("matrix_spb", 0)
        gpu_ctx.set_option("matrix_fused", -1)
    assert (tot_g == tot).all() and (ps_g == ps).all()
    for b in (A, Bt, F):
        if b is not None:
            b.free()


@pytest.mark.parametrize("f_kind", ["cluster", "run", "bitmap", "none"])
def test_count_matrix_fused_window_edges(gpu_ctx, oracle, B, f_kind, shadow_mode):
    """The in-kernel decode works through a container in eight windows of 8192 values, from work lists
    built out of the batch's window index: arrays clustered inside ONE window (up to 32 items of one
    row in one stage, more than a thousand items per stage: the list loop past the prefetched items),
    values on both sides of every window edge, runs that straddle edges or cover several windows,
    a window without any value, the filter itself an array / a run list, tile edges (40 x 33 rows)."""
    O = oracle
    rng = D.rng_for(58)
    n_shards, n_a, n_b = 2, 40, 33

    def container(kind):
        if kind == "cluster":
            w = int(rng.integers(0, 8))
            n = int(rng.choice([1, 7, 8, 9, 127, 128, 129, 1000, 4095]))
            return O.OContainer.array(np.sort(rng.choice(8192, size=n, replace=False)) + w * 8192)
        if kind == "edges":
            v = np.array([0] + [e for w in range(1, 8) for e in (w * 8192 - 1, w * 8192)] + [65535])
            return O.OContainer.array(v[rng.random(v.size) < 0.8])
        if kind == "run_straddle":
            return O.OContainer.run([(8000, 8400), (16383, 16384), (24576, 24576), (30000, 50000), (57343, 57343), (65535, 65535)])
        if kind == "run_cluster":  # 300 short runs inside one window
            w = int(rng.integers(0, 8))
            st = w * 8192 + np.arange(300) * 27 + rng.integers(0, 5, 300)
            return O.OContainer.run([(int(a), int(a) + int(rng.integers(0, 20))) for a in st])
        if kind == "two_windows":  # nothing in six of the eight windows
            return O.OContainer.array(np.concatenate([np.arange(8192 * 2 + 5, 8192 * 2 + 700, 3), np.arange(8192 * 6, 8192 * 6 + 40)]))
        return D.oracle_container(rng, D.KINDS[int(rng.integers(0, len(D.KINDS)))])

    kinds = ["cluster", "cluster", "edges", "run_straddle", "run_cluster", "two_windows", "any", "any"]

    def row():
        return {s: container(kinds[int(rng.integers(0, len(kinds)))]) for s in range(16) if rng.random() > 0.1}

    a_rows = [[row() for _ in range(n_a)] for _ in range(n_shards)]
    b_rows = [[row() for _ in range(n_b)] for _ in range(n_shards)]
    # one slot where EVERY row is a 4095-value array inside window 3: 73 x 32 items in one stage
    for s in range(n_shards):
        for r in a_rows[s] + b_rows[s]:
            r[5] = O.OContainer.array(np.sort(rng.choice(8192, size=4095, replace=False)) + 3 * 8192)
    f_rows = None
    if f_kind != "none":
        f_rows = []
        for s in range(n_shards):
            fr = {}
            for k in range(16):
                if f_kind == "cluster":
                    fr[k] = container("cluster") if k != 5 else O.OContainer.array(np.arange(3 * 8192, 4 * 8192, 2))
                elif f_kind == "run":
                    fr[k] = container("run_straddle" if k % 2 else "run_cluster")
                else:
                    fr[k] = O.OContainer.bitmap(D.words_of(D.vals_density(rng, 0.5)))
            f_rows.append(fr)

    def obm(r):
        return O.OBitmap.from_containers(list(r.items()))

    A = gpu_ctx.upload([D.to_fbk_row(r) for s in a_rows for r in s])
    Bt = gpu_ctx.upload([D.to_fbk_row(r) for s in b_rows for r in s])
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in f_rows]) if f_rows else None
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards) if F is not None else None
    try:
        gpu_ctx.set_option("matrix_fused", 1)
        tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
        gpu_ctx.set_option("matrix_fused", 0)
        tot_g, ps_g = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
    finally:
        gpu_ctx.set_option("matrix_fused", -1)
    for s in range(n_shards):
        e = B.groupby_counts(B.Fragment([obm(r) for r in a_rows[s]]), B.Fragment([obm(r) for r in b_rows[s]]), obm(f_rows[s]) if f_rows else None)
        assert (ps[s] == e).all(), (s, np.argwhere(ps[s] != e)[:5])
    assert (ps_g == ps).all() and (tot_g == tot).all()
    for b in (A, Bt, F):
        if b is not None:
            b.free()


def test_count_matrix_window_index_follows_rewritten_batches(gpu_ctx, oracle, B, shadow_mode):
    """The in-kernel decode reads a per-batch window index that is built on first use.  A plan's output batch
    is rewritten by every run of the plan (8 KiB cells, a different set of them nil each time): used as the
    GroupBy filter after each run, the index is dropped with the rewrite and rebuilt on the next use."""
    O = oracle
    rng = D.rng_for(59)
    n_shards, n_a, n_b = 3, 34, 33

    def obm(r):
        return O.OBitmap.from_containers(list(r.items()))

    a_rows = [[D.random_row(rng, 0, p_missing=0.2) for _ in range(n_a)] for _ in range(n_shards)]
    b_rows = [[D.random_row(rng, 0, p_missing=0.2) for _ in range(n_b)] for _ in range(n_shards)]
    x_rows = [D.random_row(rng, 0, p_missing=0.1) for _ in range(n_shards)]
    y_rows = [D.random_row(rng, 0, p_missing=0.1) for _ in range(n_shards)]
    A = gpu_ctx.upload([D.to_fbk_row(r) for s in a_rows for r in s])
    Bt = gpu_ctx.upload([D.to_fbk_row(r) for s in b_rows for r in s])
    X = gpu_ctx.upload([D.to_fbk_row(r) for r in x_rows])
    Y = gpu_ctx.upload([D.to_fbk_row(r) for r in y_rows])
    idx = np.arange(n_shards)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    plan = gpu_ctx.plan(X, idx, Y, idx)
    try:
        gpu_ctx.set_option("matrix_fused", 1)
        for op, name in ((L.OP_AND, "intersect"), (L.OP_XOR, "xor"), (L.OP_ANDNOT, "difference"), (L.OP_AND, "intersect")):
            plan.setop(op)
            _, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, plan.output(), idx, per_shard=True)
            for s in range(n_shards):
                f = getattr(obm(x_rows[s]), name)(obm(y_rows[s]))
                e = B.groupby_counts(B.Fragment([obm(r) for r in a_rows[s]]), B.Fragment([obm(r) for r in b_rows[s]]), f)
                assert (ps[s] == e).all(), (name, s)
    finally:
        gpu_ctx.set_option("matrix_fused", -1)
        plan.free()
    for b in (A, Bt, X, Y):
        b.free()


def test_time_kernels_option_reports_the_dominant_kernel(gpu_ctx):
    """Option time_kernels: after a query-level call, last_kernel_ns is the duration of its dominant kernel
    (HIP events recorded by the library around the launch) — what bench.py reports as kernel_us."""
    import time

    w = D.dense_rows(2 * 40, 0.4, 77)
    A = gpu_ctx.upload_dense(w)
    rows = np.arange(80).reshape(2, 40)
    try:
        gpu_ctx.set_option("time_kernels", 1)
        t0 = time.perf_counter()
        gpu_ctx.count_matrix(A, rows[:, :20], A, rows[:, 20:])
        wall_ns = (time.perf_counter() - t0) * 1e9
        k1 = gpu_ctx.get_option("last_kernel_ns")
        assert 0 < k1 < wall_ns
        assert gpu_ctx.get_option("last_kernel_ns") == k1  # stays until the next timed call
        gpu_ctx.union_n_intersection_count(A, rows, A, np.zeros(2, dtype=np.uint32))
        assert gpu_ctx.get_option("last_kernel_ns") > 0
    finally:
        gpu_ctx.set_option("time_kernels", 0)
    A.free()


def test_rows_vs_filter_many_rows(gpu_ctx, oracle):
    """TopK over a field with more rows than the matrix limit (5000 sparse rows x 2 shards)."""
    O = oracle
    rng = D.rng_for(61)
    n_shards, n_a = 2, 5000
    rows = []
    for s in range(n_shards):
        for i in range(n_a):
            slot = int(rng.integers(0, 16))
            vals = np.sort(rng.choice(65536, size=int(rng.integers(1, 40)), replace=False))
            rows.append({s * 16 + slot: O.OContainer.array(vals)})
    filt = [D.random_row(rng, s) for s in range(n_shards)]
    A = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in filt])
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    tot, ps = gpu_ctx.count_matrix(A, ra, F, np.arange(n_shards).reshape(-1, 1), per_shard=True)
    for s in range(n_shards):
        for i in range(0, n_a, 37):
            r = rows[s * n_a + i]
            exp = sum(O.intersection_count(c, filt[s][k]) for k, c in r.items() if k in filt[s])
            assert int(ps[s, i, 0]) == exp, (s, i)
    assert int(tot.sum()) == int(ps.sum())
    with pytest.raises(L.FbkError):
        gpu_ctx.count_matrix(A, ra, A, ra)  # a 5000 x 5000 matrix is refused
    A.free()
    F.free()


def test_bsi_add_is_integer_addition(gpu_ctx, oracle):
    """TestAdd (roaring/add_test.go:311-407) on the device adder: random position -> count maps
    (heavy-tailed, up to 2^44), bit-sliced into planes of any encoding; the planes of
    fbk_bsi_add decode to x + y at every position.  Operands of different depth, a group where
    one operand is empty, and carries into plane D."""
    O = oracle
    rng = D.rng_for(67)
    n_groups = 4

    def planes_of(vals, depth):
        rows = []
        for i in range(depth):
            bm = O.bitmap_from_values([p for p, v in vals.items() if (v >> i) & 1])
            rows.append({k: D.to_fbk(c) for k, c in bm.items() if c.n})
        return rows

    xs, ys = [], []
    for g in range(n_groups):
        n = [3000, 40, 70000, 500][g]
        pos = rng.choice(1 << 20, size=n, replace=False)
        heavy = lambda m: (rng.pareto(1.2, size=m) * 7).astype(np.uint64) % (1 << 44)  # noqa: E731
        xv = {int(p): int(v) for p, v in zip(pos, heavy(n)) if v}
        yv = {int(p): int(v) for p, v in zip(pos[: n // 2], heavy(n // 2)) if v}
        extra = rng.choice(1 << 20, size=n // 3 + 1, replace=False)
        for p, v in zip(extra, heavy(len(extra))):
            if v:
                yv[int(p)] = int(v)
        if g == 1:
            yv = {}  # "there are no values in y" (bsi.go:158)
        if g == 3:
            xv.update({5: (1 << 44) - 1, 6: (1 << 44) - 1})  # carries ripple through every plane
            yv.update({5: 1, 6: (1 << 44) - 1})
        xs.append(xv)
        ys.append(yv)
    dx, dy = 44, 44
    for depths in ((44, 44), (44, 20), (7, 44)):
        dx, dy = depths
        xm = [{p: v & ((1 << dx) - 1) for p, v in xv.items() if v & ((1 << dx) - 1)} for xv in xs]
        ym = [{p: v & ((1 << dy) - 1) for p, v in yv.items() if v & ((1 << dy) - 1)} for yv in ys]
        xrows = [r for xv in xm for r in planes_of(xv, dx)]
        yrows = [r for yv in ym for r in planes_of(yv, dy)]
        X, Y = gpu_ctx.upload(xrows), gpu_ctx.upload(yrows)
        rx = np.arange(n_groups * dx).reshape(n_groups, dx)
        ry = np.arange(n_groups * dy).reshape(n_groups, dy)
        for flags in (0, L.SETOP_OPTIMIZE):
            out = gpu_ctx.bsi_add(X, rx, Y, ry, flags)
            rows = out.download()
            Dp = max(dx, dy) + 1
            assert len(rows) == n_groups * Dp
            for g in range(n_groups):
                got = {}
                for i in range(Dp):
                    for k, c in rows[g * Dp + i].items():
                        bits = np.unpackbits(c.words().view(np.uint8), bitorder="little")
                        for v in np.nonzero(bits)[0]:
                            p = ((k & 15) << 16) + int(v)
                            got[p] = got.get(p, 0) | (1 << i)
                exp = dict(xm[g])
                for p, v in ym[g].items():
                    exp[p] = exp.get(p, 0) + v
                assert got == exp, (depths, flags, g)
            out.free()
        X.free()
        Y.free()


def test_bsi_distinct_vs_definition(gpu_ctx, B):
    """fbk_bsi_distinct against the definition executeDistinctShardBSI implements
    (executor.go:2034-2153: the value of every column of exists ∩ filter, then the set union over
    shards): mixed-encoding planes (densify path) and dense planes, with and without a filter,
    depth 64 with magnitude bit 63 (int64 wrap), many ties, an empty shard."""
    rng = D.rng_for(71)
    for depth in (5, 20, 64):
        frags, filts, vals_all = [], [], []
        for s in range(5):
            ncol = [30000, 700, 1 << 16, 0, 11][s]
            cols = rng.choice(1 << 20, size=ncol, replace=False)
            hi = 1 << min(depth, 62)
            mag = rng.integers(0, hi, size=ncol)
            if s in (0, 2):
                mag = mag % 1000  # many repeated values
            sign = np.where(rng.random(ncol) < 0.4, -1, 1)
            vals = {int(c): int(m) * int(g) for c, m, g in zip(cols, mag, sign)}
            if depth == 64 and s == 1:
                vals[int(cols[0])] = (1 << 63) + 9  # wraps to a negative int64 as in the reference
                vals[int(cols[1])] = -((1 << 63) + 9)
            vals_all.append(vals)
            frags.append(B.bsi_fragment_from_values(vals, depth))
            filts.append(B.row_from_columns([int(c) for c in cols[:: 2 + s]] + [5, 70000]))
        batch, base = upload_bsi(gpu_ctx, frags)
        F = gpu_ctx.upload([fbk_row_of_bitmap(f) for f in filts])

        def wrap(v):
            v &= (1 << 64) - 1
            return v - (1 << 64) if v >= (1 << 63) else v

        exp_all = sorted({wrap(v) for vals in vals_all for v in vals.values()})
        got = gpu_ctx.bsi_distinct(batch, base, depth)
        assert got.tolist() == exp_all, depth
        exp_f = set()
        for s, vals in enumerate(vals_all):
            fs = set(B.columns(filts[s]))
            exp_f |= {wrap(v) for c, v in vals.items() if c in fs}
        got = gpu_ctx.bsi_distinct(batch, base, depth, F, np.arange(5))
        assert got.tolist() == sorted(exp_f), depth
        batch.free()
        F.free()
    # dense planes (no densify pass): config 5's layout, values checked through their count
    n_shards, depth = 3, 12
    w = D.dense_rows(n_shards * (depth + 2), 0.5, 901).reshape(n_shards, depth + 2, 16, 1024)
    w[:, 1:] &= w[:, :1]
    batch = gpu_ctx.upload_dense(w.reshape(-1))
    base = np.arange(n_shards, dtype=np.uint32) * (depth + 2)
    got = gpu_ctx.bsi_distinct(batch, base, depth)
    bits = np.unpackbits(w.reshape(n_shards, depth + 2, -1).view(np.uint8), axis=2, bitorder="little").astype(np.int64)
    mag = sum(bits[:, 2 + i] << i for i in range(depth))
    val = np.where(bits[:, 1] == 1, -mag, mag)
    exp = np.unique(val[bits[:, 0] == 1])
    assert got.tolist() == exp.tolist()
    batch.free()


@pytest.mark.parametrize("flags", [0, L.SETOP_OPTIMIZE])
def test_shift_vs_oracle(gpu_ctx, oracle, flags):
    """fbk_shift against the restated Row.Shift (oracle/pyshift.py, pinned to the reference's own
    vectors in tests/test_oracle_shift.py): rows of every encoding over shards 0, 1 and 3, the last
    column of several containers and shards set so that carries cross container AND shard
    boundaries — into an existing shard (1), into a shard that exists only because of the carried
    bit (2 and 4) — then shifted three times."""
    from oracle import pyshift as S

    O = oracle
    rng = D.rng_for(61)
    SW = 1 << 20
    shards = [0, 1, 3]
    segs = {}
    for sh in shards:
        row = D.random_row(rng, sh, p_missing=0.2)
        for slot in (2, 7, 15):  # force value 65535 (carry out) in every encoding
            key = sh * 16 + slot
            words = row[key].words() if key in row else np.zeros(1024, dtype=np.uint64)
            words[1023] |= np.uint64(1) << np.uint64(63)
            typ = [O.ARRAY, O.BITMAP, O.RUN][slot % 3]
            c = O.OContainer.from_words(words, typ) if int(np.bitwise_count(words).sum()) < 4096 or typ != O.ARRAY else O.OContainer.bitmap(words)
            row[key] = c
        if sh == 1:
            row.pop(sh * 16 + 0, None)  # the carry from shard 0 lands in a nil container
        segs[sh] = sorted(row.items())
    want = segs
    have = {sh: {k: c for k, c in items} for sh, items in segs.items()}  # shard -> {key: oracle container}
    for step in range(3):
        want = S.row_shift(want, 1)
        exp_cols = S.row_columns(want)
        # one device call per step: every shard that has a row or a predecessor with a row
        out_shards = sorted(set(have) | {s + 1 for s in have})
        order = sorted(have)
        batch = gpu_ctx.upload([D.to_fbk_row(have[s]) for s in order])
        idx = {s: i for i, s in enumerate(order)}
        rows = [idx.get(s, gpu_ctx.NO_ROW) for s in out_shards]
        carry = [idx.get(s - 1, gpu_ctx.NO_ROW) for s in out_shards]
        out, counts = gpu_ctx.shift(batch, rows, carry, flags)
        got_rows = out.download()
        got_cols, new_have = [], {}
        for i, s in enumerate(out_shards):
            n = 0
            for key, c in got_rows[i].items():
                assert key >> 4 == s, (key, s)  # output keys carry the shard of the row
                vals = np.nonzero(np.unpackbits(c.words().view(np.uint8), bitorder="little"))[0]
                got_cols.extend(((key << 16) + vals).tolist())
                n += vals.size
            assert n == int(counts[i])
            if n:
                new_have[s] = {k: O.OContainer.from_words(c.words(), O.BITMAP) for k, c in got_rows[i].items()}
        assert sorted(got_cols) == exp_cols, step
        if flags & L.SETOP_OPTIMIZE:  # Container.optimize() encodings on the way out
            for i in range(len(out_shards)):
                for key, c in got_rows[i].items():
                    assert c.typ == O.optimize(O.OContainer.bitmap(c.words())).typ
        have = new_have
        batch.free()
        out.free()
    with pytest.raises(L.FbkError):
        b = gpu_ctx.upload([D.to_fbk_row(D.random_row(rng, 0))])
        try:
            gpu_ctx.shift(b, [5])
        finally:
            b.free()


def test_chunk_ring_boundaries(gpu_ctx, oracle):
    """The fold and TopK kernels consume payloads as 1 KiB chunks (512 array values / 256 runs /
    128 bitmap words): arrays and run lists whose lengths sit on, just before and just after every
    chunk edge, singletons, the largest policy sizes (4095 values, 2048 runs) and the first sizes
    beyond them — one group per shape so that a miscounted chunk cannot cancel out — through
    fold OR / XOR / ANDNOT (with counts), the fused |∪ ∩ F| and the rows-vs-filter counts."""
    O = oracle
    rng = D.rng_for(67)
    array_sizes = [1, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 3071, 3072, 3073, 4095, 4096, 4097, 6000]
    run_counts = [1, 3, 4, 5, 255, 256, 257, 511, 512, 513, 1024, 2047, 2048, 2049, 3000]
    shapes = []
    for n in array_sizes:
        shapes.append(O.OContainer.array(np.sort(rng.choice(65536, size=n, replace=False))))
    for r in run_counts:
        starts = np.sort(rng.choice(65536 // 8, size=r, replace=False)) * 8  # runs of 1..6 values, never adjacent
        shapes.append(O.OContainer.run([(int(s), int(s + rng.integers(0, 6))) for s in starts]))
    shapes.append(O.OContainer.run([(0, 65535)]))
    shapes.append(O.OContainer.bitmap(rng.integers(0, 2**64, 1024, dtype=np.uint64)))
    # group g = [shape g, a bitmap row, shape (g+1) % n]: slot = g % 16 of shard row 0
    rows, groups = [], []
    partner = O.OContainer.bitmap(rng.integers(0, 2**64, 1024, dtype=np.uint64) & rng.integers(0, 2**64, 1024, dtype=np.uint64))
    for g, c in enumerate(shapes):
        slot = g % 16
        ids = []
        for cc in (c, partner, shapes[(g + 1) % len(shapes)]):
            ids.append(len(rows))
            rows.append({slot: cc})
        groups.append(ids)
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    frow = {s: O.OContainer.bitmap(rng.integers(0, 2**64, 1024, dtype=np.uint64)) for s in range(16)}
    F = gpu_ctx.upload([D.to_fbk_row(frow)])
    fb = O.OBitmap.from_containers(list(frow.items()))
    obm = [O.OBitmap.from_containers(list(r.items())) for r in rows]
    for op in (L.OP_OR, L.OP_XOR, L.OP_ANDNOT):
        out, cnt = gpu_ctx.fold_n(op, batch, groups)
        res = out.download()
        for g, ids in enumerate(groups):
            bms = [obm[i] for i in ids]
            exp = bms[0].union(*bms[1:]) if op == L.OP_OR else bms[0].difference(*bms[1:]) if op == L.OP_ANDNOT else bms[0].xor(bms[1]).xor(bms[2])
            assert int(cnt[g]) == exp.count(), (op, g)
            assert (row_words(res[g]) == bitmap_words(exp)).all(), (op, g)
        out.free()
    fused = gpu_ctx.union_n_intersection_count(batch, groups, F, np.zeros(len(groups), dtype=np.uint32))
    for g, ids in enumerate(groups):
        bms = [obm[i] for i in ids]
        assert int(fused[g]) == bms[0].union(*bms[1:]).intersection_count(fb), g
    # rows vs filter: every row of the batch against the filter row (one "shard" of len(rows) rows)
    tot = gpu_ctx.count_matrix(batch, np.arange(len(rows)).reshape(1, -1), F, np.zeros((1, 1), dtype=np.uint32))
    for i in range(len(rows)):
        assert int(tot[i, 0]) == obm[i].intersection_count(fb), i
    batch.free()
    F.free()


@pytest.mark.parametrize("use_filter", [True, False])
def test_topk_on_device_vs_oracle(gpu_ctx, oracle, use_filter):
    """fbk_topk (counts, reduce over shards and the PivotDescending order on the device) against
    oracle intersection counts ordered on the host: ties (duplicated rows) keep ascending row index,
    rows without a common column are dropped, k = 0 / 1 / 7 / more than there are, several passes
    over the shards, and the FBK_E_CAPACITY report."""
    O = oracle
    rng = D.rng_for(71)
    n_shards, n_a = 3, 90
    rows, ra = [], []
    for s in range(n_shards):
        ids = []
        base = [D.random_row(rng, s, p_missing=0.5) for _ in range(n_a)]
        for i in range(n_a):
            if i % 10 == 9:
                base[i] = base[i - 1]  # a tie: the same columns as the previous row
            if i % 13 == 0:
                base[i] = {}  # a row with no columns at all
            ids.append(len(rows))
            rows.append(base[i])
        ra.append(ids)
    frows = [D.random_row(rng, s, p_missing=0.3) for s in range(n_shards)]
    A = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])
    F = gpu_ctx.upload([D.to_fbk_row(r) for r in frows])
    ra = np.array(ra, dtype=np.uint32)
    tot = np.zeros(n_a, dtype=np.int64)
    for s in range(n_shards):
        fb = O.OBitmap.from_containers(list(frows[s].items()))
        for i in range(n_a):
            bm = O.OBitmap.from_containers(list(rows[ra[s, i]].items()))
            tot[i] += bm.intersection_count(fb) if use_filter else bm.count()
    order = sorted([i for i in range(n_a) if tot[i]], key=lambda i: (-tot[i], i))
    assert len(order) < n_a and len(set(tot[order].tolist())) < len(order)  # zeros and ties are present
    fargs = (F, np.arange(n_shards)) if use_filter else (None, None)
    try:
        for mode in ("0", "1"):  # ordered on the host (small fields) / by the device radix sort
            gpu_ctx.set_option("topk_device_sort", int(mode))
            for k in (0, 1, 7, 1000):
                idx, cnt = gpu_ctx.topk(A, ra, k, *fargs)
                exp = order if k == 0 else order[:k]
                assert idx.tolist() == exp and cnt.tolist() == tot[exp].tolist(), (mode, k)
    finally:
        gpu_ctx.set_option("topk_device_sort", -1)
    try:  # one shard per pass
        gpu_ctx.set_option("matrix_pass_kb", int("1"))
        idx, cnt = gpu_ctx.topk(A, ra, 5, *fargs)
        assert idx.tolist() == order[:5] and cnt.tolist() == tot[order[:5]].tolist()
    finally:
        gpu_ctx.set_option("matrix_pass_kb", 1048576)
    # a buffer smaller than the result: FBK_E_CAPACITY and the needed size
    import ctypes as C

    small_i, small_c, n = np.zeros(2, np.uint32), np.zeros(2, np.uint64), C.c_uint32()
    need = len([i for i in range(n_a) if any(rows[ra[s, i]] for s in range(n_shards))])
    for mode in ("0", "1"):
        gpu_ctx.set_option("topk_device_sort", int(mode))
        try:
            rc = gpu_ctx.lib.fbk_topk(gpu_ctx.h, A.h, ra.ctypes.data, n_a, None, None, n_shards, 0, small_i.ctypes.data, small_c.ctypes.data, 2, C.byref(n))
        finally:
            gpu_ctx.set_option("topk_device_sort", -1)
        assert rc == L.FBK_E_CAPACITY and n.value == need
    A.free()
    F.free()


def test_fragment_topn_vectors_through_fbk_topk(gpu_ctx, oracle):
    """TestFragment_TopN_Intersect / _Intersect_Large / _IDs (fragment_internal_test.go:1174-1273),
    the reference's own TopN known answers, through fbk_topk (one shard; both ordering paths;
    the RowIDs form passes the chosen rows — a missing id is an empty row — and no filter)."""
    from oracle import pybsi as PB
    from test_oracle_bsi import topn_vectors

    for rows, src, n, want, row_ids in topn_vectors():
        ids = sorted(rows) if row_ids is None else row_ids
        batch = gpu_ctx.upload([fbk_row_of_bitmap(PB.row_from_columns(rows[i])) if rows.get(i) else {} for i in ids])
        F = gpu_ctx.upload([fbk_row_of_bitmap(PB.row_from_columns(src))]) if src is not None else None
        fargs = (F, np.zeros(1, dtype=np.uint32)) if F is not None else (None, None)
        try:
            for mode in ("0", "1"):
                gpu_ctx.set_option("topk_device_sort", int(mode))
                idx, cnt = gpu_ctx.topk(batch, np.arange(len(ids)).reshape(1, -1), n, *fargs)
                assert [(ids[i], int(c)) for i, c in zip(idx.tolist(), cnt.tolist())] == want, mode
        finally:
            gpu_ctx.set_option("topk_device_sort", -1)
        batch.free()
        if F is not None:
            F.free()


def test_bsi_add_all_bitmap_planes_fast_path(gpu_ctx, oracle):
    """k_bsi_add's round-6 path for operands whose planes are all bitmaps (or nil): descriptors staged in the LDS, four planes per
    round of loads, cardinalities and run counts left per wave and added up once at the end.  Dense planes (every column holds a
    value), operands of different depth, a nil plane in the middle, and planes that are long RUNS across wave and half-container
    boundaries (words 127 / 128, 511 / 512 ...): x + y per column against numpy, and with FBK_SETOP_OPTIMIZE every result
    container in Container.optimize()'s encoding (the run counts the kernel hands to the encoder are what decides it)."""
    O = oracle
    rng = D.rng_for(6061)
    n_groups, ncol = 2, 1 << 20
    for dx, dy in ((12, 12), (9, 13), (5, 2)):
        xv = rng.integers(0, 1 << dx, size=(n_groups, ncol), dtype=np.uint64)
        yv = rng.integers(0, 1 << dy, size=(n_groups, ncol), dtype=np.uint64)
        # group 1: long constant stretches, so that planes are runs whose ends fall on the boundaries between waves' words
        edges = [0, 64 * 127 + 63, 64 * 128, 64 * 511 + 1, 64 * 512, 64 * 640 - 1, 65536, 65536 + 64 * 256, 3 * 65536 - 5, 5 * 65536 + 64 * 384 + 63, ncol]
        for a, b in zip(edges[:-1], edges[1:]):
            xv[1, a:b] = rng.integers(0, 1 << dx)
            yv[1, a:b] = rng.integers(0, 1 << dy)
        xv[:, :] &= ~np.uint64(1 << 2)  # plane 2 of x is nil in every group
        def planes(v, depth):
            out = np.zeros((n_groups * depth, 16, 1024), dtype=np.uint64)
            for g in range(n_groups):
                for i in range(depth):
                    bits = ((v[g] >> np.uint64(i)) & np.uint64(1)).astype(np.uint8)
                    out[g * depth + i] = np.packbits(bits, bitorder="little").view(np.uint64).reshape(16, 1024)
            return out
        X, Y = gpu_ctx.upload_dense(planes(xv, dx)), gpu_ctx.upload_dense(planes(yv, dy))
        rx = np.arange(n_groups * dx).reshape(n_groups, dx)
        ry = np.arange(n_groups * dy).reshape(n_groups, dy)
        zv = xv + yv
        Dp = max(dx, dy) + 1
        for flags in (0, L.SETOP_OPTIMIZE):
            out = gpu_ctx.bsi_add(X, rx, Y, ry, flags)
            rows = out.download()
            assert len(rows) == n_groups * Dp
            for g in range(n_groups):
                for i in range(Dp):
                    bits = ((zv[g] >> np.uint64(i)) & np.uint64(1)).astype(np.uint8)
                    exp_w = np.packbits(bits, bitorder="little").view(np.uint64).reshape(16, 1024)
                    got = rows[g * Dp + i]
                    exp_bm = {s: O.OContainer.bitmap(exp_w[s]) for s in range(16) if exp_w[s].any()}
                    assert {k & 15 for k in got} == set(exp_bm), (dx, dy, flags, g, i)
                    for k, c in got.items():
                        assert np.array_equal(c.words(), exp_w[k & 15]), (dx, dy, flags, g, i, k)
                    if flags & L.SETOP_OPTIMIZE:
                        assert_optimized_like_oracle(O, got, exp_bm)
            out.free()
        X.free()
        Y.free()


def test_bsi_add_reference_cases(gpu_ctx, oracle):
    """TestBSIAddCases (bsi_test.go:116-160): the reference's own AddBSI inputs — twenty positions
    with counts up to 9 023 592 401, and the two-position case — through fbk_bsi_add; every
    position decodes to a + b."""
    O = oracle
    import json
    import os

    lit = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "literal_vectors.json")))  # extracted mechanically from bsi_test.go
    cases = [(c["positions"], c["a"], c["b"]) for c in lit["bsi_add_cases"]]
    depth = 34  # 9023592401 < 2^34

    def planes_of(vals):
        rows = []
        for i in range(depth):
            bm = O.bitmap_from_values([p for p, v in vals.items() if (v >> i) & 1])
            rows.append({k: D.to_fbk(c) for k, c in bm.items() if c.n})
        return rows

    xrows, yrows = [], []
    for pos, a, b in cases:
        xrows += planes_of(dict(zip(pos, a)))
        yrows += planes_of(dict(zip(pos, b)))
    X, Y = gpu_ctx.upload(xrows), gpu_ctx.upload(yrows)
    r = np.arange(len(cases) * depth).reshape(len(cases), depth)
    out = gpu_ctx.bsi_add(X, r, Y, r, L.SETOP_OPTIMIZE)
    rows = out.download()
    for g, (pos, a, b) in enumerate(cases):
        got = {}
        for i in range(depth + 1):
            for k, c in rows[g * (depth + 1) + i].items():
                bits = np.unpackbits(c.words().view(np.uint8), bitorder="little")
                for v in np.nonzero(bits)[0]:
                    p = ((k & 15) << 16) + int(v)
                    got[p] = got.get(p, 0) | (1 << i)
        assert got == {p: x + y for p, x, y in zip(pos, a, b) if x + y}, g
    for bt in (out, X, Y):
        bt.free()


def test_fold_optimize_in_the_epilogue(gpu_ctx, oracle):
    """Union / Xor / Difference / Intersect of k rows + optimize() in the fold kernel's own epilogue: the bits and counts of the
    oracle's fold, every container in optimize()'s encoding of the oracle's result (roaring.go:3412-3461), the call repeatable
    byte for byte (descriptors, payload, Pilosa-roaring image).  (Until round 5 a separate re-encode pass over 8 KiB cells was
    kept behind option fold_encode = 0 and compared byte for byte; the pair set-ops still run it: setop_direct_encode < 2.)  Rows are chosen to land on every outcome: nil, short / long arrays (incl. 4095 values), run lists of
    1 .. 2048 intervals (incl. runs spanning the two halves of a cell and runs ending at 65535), bitmaps, full containers."""
    O = oracle
    rng = D.rng_for(4242)

    def cont(kind):
        if kind == "full":
            return O.OContainer.run([(0, 65535)])
        if kind == "runs":
            nr = int(rng.choice([1, 2, 7, 64, 700, 2048]))
            per = 65536 // nr
            st = np.arange(nr) * per + rng.integers(0, max(1, per // 3), nr)
            ln = rng.integers(1, max(2, per // 2), nr)
            return O.OContainer.run([(int(s), int(min(s + l, 65535))) for s, l in zip(st, ln)])
        if kind == "edge":  # a run across the middle of the cell and one up to the last value
            return O.OContainer.run([(32700, 32900), (65500, 65535)])
        if kind == "arr":
            n = int(rng.choice([1, 5, 300, 2047, 4095]))
            return O.OContainer.array(np.sort(rng.choice(65536, n, replace=False)))
        return O.OContainer.bitmap(D.words_of(np.sort(rng.choice(65536, int(rng.choice([5000, 30000, 65000])), replace=False))))

    kinds = ["full", "runs", "edge", "arr", "bm", "runs", "arr"]
    n_groups, k = 12, 5
    rows, groups = [], []
    for g in range(n_groups):
        ids = []
        for _ in range(k):
            row = {}
            for s in range(16):
                if rng.random() < 0.25:
                    continue
                # groups 0-3: sparse operands only (array / run results); others: everything
                kind = kinds[int(rng.integers(3 if g < 4 else 0, len(kinds)))] if g >= 4 else ["arr", "runs", "edge"][int(rng.integers(0, 3))]
                row[g * 16 + s] = cont(kind)
            ids.append(len(rows))
            rows.append(row)
        groups.append(ids)
    groups = np.array(groups, dtype=np.uint32)
    batch = gpu_ctx.upload([D.to_fbk_row(r) for r in rows])

    def fold(op, bms):
        if op == L.OP_OR:
            return bms[0].union(*bms[1:])
        if op == L.OP_ANDNOT:
            return bms[0].difference(*bms[1:])
        acc = bms[0]
        for b in bms[1:]:
            acc = acc.xor(b) if op == L.OP_XOR else acc.intersect(b)
        return acc

    for op in (L.OP_OR, L.OP_XOR, L.OP_ANDNOT, L.OP_AND):
        got = {}
        for rep in (1, 0):
            out, cnt = gpu_ctx.fold_n(op, batch, groups, L.SETOP_OPTIMIZE)
            d, p, n_rows = out.download_flat()
            got[rep] = (d.copy(), p.copy(), cnt.copy(), out.to_roaring(), out.download())
            out.free()
        (d1, p1, c1, img1, res1), (d0, p0, c0, img0, _) = got[1], got[0]
        assert (c1 == c0).all() and d1.tobytes() == d0.tobytes() and p1.tobytes() == p0.tobytes() and img1 == img0, op
        types = set()
        for g, ids in enumerate(groups):
            exp = fold(op, [O.OBitmap.from_containers(list(rows[i].items())) for i in ids])
            assert int(c1[g]) == exp.count(), (op, g)
            assert (row_words(res1[g]) == bitmap_words(exp)).all(), (op, g)
            assert_optimized_like_oracle(O, res1[g], exp)
            types |= {c.typ for c in res1[g].values()}
        if op != L.OP_AND:  # (the intersection of five rows is sparse)
            assert types >= {L.TYPE_ARRAY, L.TYPE_BITMAP, L.TYPE_RUN}, (op, types)  # every encoding was produced
    batch.free()


def test_setop_optimize_inside_the_kernel_equals_the_separate_pass(gpu_ctx, oracle):
    """Pair set-ops + optimize(): option setop_direct_encode = 2 (round 4: Container.optimize() applied by the set-op kernel,
    the encoded container written into the head of its cell), 1 (round 2: small results as arrays, then the re-encode pass) and
    0 (bitmap cells, then the re-encode pass) give the same descriptors, payload bytes and roaring image, for both generations
    of the pair kernels; every container has optimize()'s encoding of the oracle's result.  Also as a PLAN: launch-only with
    mode 2, refused with the others."""
    O = oracle
    rng = D.rng_for(4343)

    def cont():
        k = int(rng.integers(0, 8))
        if k == 0:
            return O.OContainer.run([(0, 65535)])
        if k == 1:
            nr = int(rng.choice([1, 3, 40, 900, 2048]))
            per = 65536 // nr
            st = np.arange(nr) * per + rng.integers(0, max(1, per // 3), nr)
            ln = rng.integers(1, max(2, per // 2), nr)
            return O.OContainer.run([(int(s), int(min(s + l, 65535))) for s, l in zip(st, ln)])
        if k == 2:
            return O.OContainer.run([(32700, 32900), (65500, 65535)])
        if k in (3, 4):
            return O.OContainer.array(np.sort(rng.choice(65536, int(rng.choice([1, 7, 60, 300, 2047, 4095])), replace=False)))
        if k == 5:  # consecutive values: optimize() turns the intersection of arrays into runs
            s0 = int(rng.integers(0, 60000))
            return O.OContainer.array(np.arange(s0, s0 + int(rng.integers(2, 50))))
        return O.OContainer.bitmap(D.words_of(np.sort(rng.choice(65536, int(rng.choice([5000, 30000, 65000])), replace=False))))

    n = 40
    rows_a = [{r * 16 + s: cont() for s in range(16) if rng.random() > 0.15} for r in range(n)]
    rows_b = [{r * 16 + s: cont() for s in range(16) if rng.random() > 0.15} for r in range(n)]
    A, Bt = gpu_ctx.upload([D.to_fbk_row(r) for r in rows_a]), gpu_ctx.upload([D.to_fbk_row(r) for r in rows_b])
    idx = np.arange(n)
    try:
        for pk in (1, 2):
            gpu_ctx.set_option("pair_kernels", pk)
            for op, name in [(L.OP_AND, "intersect"), (L.OP_OR, "union"), (L.OP_XOR, "xor"), (L.OP_ANDNOT, "difference")]:
                got = {}
                # (mode 2: Intersect / Difference of an array operand go by table + probe, survivors written as the array they are;
                # modes 1 and 0 decode both operands into fragments and re-encode in a separate pass)
                for mode in (2, 1, 0):
                    gpu_ctx.set_option("setop_direct_encode", mode)
                    out, cnt = gpu_ctx.setop(op, A, idx, Bt, idx, flags=L.SETOP_OPTIMIZE)
                    d, p, _ = out.download_flat()
                    got[mode] = (d.tobytes(), p.tobytes(), cnt.copy(), out.to_roaring(), out.download())
                    out.free()
                for mode in (1, 0):
                    assert got[2][0] == got[mode][0] and got[2][1] == got[mode][1] and (got[2][2] == got[mode][2]).all() and got[2][3] == got[mode][3], (pk, name, mode)
                types = set()
                for r in range(n):
                    a = O.OBitmap.from_containers(list(rows_a[r].items()))
                    b = O.OBitmap.from_containers(list(rows_b[r].items()))
                    e = {"intersect": a.intersect, "union": a.union, "xor": a.xor, "difference": a.difference}[name](b)
                    assert_optimized_like_oracle(O, got[2][4][r], e)
                    assert int(got[2][2][r]) == e.count()
                    types |= {c.typ for c in got[2][4][r].values()}
                assert types == {L.TYPE_ARRAY, L.TYPE_BITMAP, L.TYPE_RUN}, (pk, name, types)
        # the plan form: launch-only with the in-kernel optimize, refused when the re-encode would be a separate pass
        gpu_ctx.set_option("pair_kernels", 0)
        gpu_ctx.set_option("setop_direct_encode", 2)
        plan = gpu_ctx.plan(A, idx, Bt, idx)
        ref, rcnt = gpu_ctx.setop(L.OP_XOR, A, idx, Bt, idx, flags=L.SETOP_OPTIMIZE)
        for _ in range(2):
            plan.setop(L.OP_XOR, L.SETOP_OPTIMIZE)
        assert (plan.read() == rcnt).all() and plan.output().to_roaring() == ref.to_roaring()
        gpu_ctx.set_option("setop_direct_encode", 1)
        with pytest.raises(L.FbkError):
            plan.setop(L.OP_XOR, L.SETOP_OPTIMIZE)
        plan.free()
        ref.free()
    finally:
        gpu_ctx.set_option("pair_kernels", 0)
        gpu_ctx.set_option("setop_direct_encode", 2)
    A.free()
    Bt.free()


def test_pair_count_arrays_on_every_row_and_half_boundary(gpu_ctx, oracle):
    """Row-pair counts whose arrays sit on both sides of every boundary the table + probe kernels know: dword rows of 128 values,
    the 32768 split of the value range, 1024 / 2048 / 4095 values, values confined to one half, consecutive values; bitmaps and
    runs mixed in.  Both generations of the pair kernels, a plan run three times, and a plan over a batch that was written on
    the device (the resolved item records follow the batch's version).  (The data set of round 4's lean array x array kernel,
    which was removed in round 5.)"""
    O = oracle
    rng = D.rng_for(4545)

    def arr(n, mode):
        if mode == 1:
            v = np.sort(rng.choice(32768, min(n, 32768), replace=False))
        elif mode == 2:
            v = np.sort(rng.choice(32768, min(n, 32768), replace=False)) + 32768
        elif mode == 3:
            s0 = int(rng.integers(32768 - n, 32768)) if n < 32768 else 0
            v = np.arange(s0, s0 + n)
        else:
            v = np.sort(rng.choice(65536, n, replace=False))
        return O.OContainer.array(v)

    def cont():
        k = int(rng.integers(0, 10))
        if k == 0:
            return O.OContainer.bitmap(D.words_of(np.sort(rng.choice(65536, int(rng.choice([5000, 40000])), replace=False))))
        if k == 1:
            nr = int(rng.choice([1, 7, 300]))
            per = 65536 // nr
            st = np.arange(nr) * per + rng.integers(0, max(1, per // 3), nr)
            ln = rng.integers(1, max(2, per // 2), nr)
            return O.OContainer.run([(int(s), int(min(s + l, 65535))) for s, l in zip(st, ln)])
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4095, int(rng.integers(1, 2049))]))
        return arr(n, int(rng.integers(0, 4)))

    n = 48
    rows_a = [{r * 16 + s: cont() for s in range(16) if rng.random() > 0.1} for r in range(n)]
    rows_b = [{r * 16 + s: cont() for s in range(16) if rng.random() > 0.1} for r in range(n)]
    for r in range(0, n, 5):  # shared values: real intersections between long arrays
        for s in range(16):
            k = r * 16 + s
            if k in rows_a[r] and k in rows_b[r] and rows_a[r][k].typ == L.TYPE_ARRAY and rows_b[r][k].typ == L.TYPE_ARRAY:
                va, vb = rows_a[r][k].values(), rows_b[r][k].values()
                rows_b[r][k] = O.OContainer.array(np.unique(np.concatenate([vb, va[::2]]))[: max(len(vb), 1)])
    A, Bt = gpu_ctx.upload([D.to_fbk_row(r) for r in rows_a]), gpu_ctx.upload([D.to_fbk_row(r) for r in rows_b])
    ia, ib = np.arange(n), (np.arange(n) * 7) % n
    ia = np.concatenate([ia, ia])
    ib = np.concatenate([ib, np.arange(n)])
    exp = [sum(O.intersection_count(rows_a[a][ka], rows_b[b][b * 16 + (ka & 15)]) for ka in rows_a[a] if b * 16 + (ka & 15) in rows_b[b]) for a, b in zip(ia, ib)]
    try:
        for pk in (2, 1, 0):
            gpu_ctx.set_option("pair_kernels", pk)
            plan = gpu_ctx.plan(A, ia, Bt, ib)
            for run in range(3):
                plan.intersection_count()
                assert plan.read().tolist() == exp, (pk, run)
            plan.free()
        # a batch written on the device (a plan's output): the plan over it resolves its items from the device descriptors
        gpu_ctx.set_option("pair_kernels", 2)
        p0 = gpu_ctx.plan(A, ia, Bt, ib)
        p0.setop(L.OP_OR)
        out = p0.output()
        m = len(ia)
        ref = gpu_ctx.intersection_count(out, np.arange(m), Bt, ib)
        # |(A ∪ B) ∩ B| = |B|
        assert ref.tolist() == [O.OBitmap.from_containers(list(rows_b[b].items())).count() for b in ib]
        plan = gpu_ctx.plan(out, np.arange(m), Bt, ib)
        for run in range(3):
            plan.intersection_count()
            assert plan.read().tolist() == ref.tolist(), run
        out.download()  # (reads the descriptors back)
        for run in range(2):
            plan.intersection_count()
            assert plan.read().tolist() == ref.tolist(), run
        plan.free()
        p0.free()
    finally:
        gpu_ctx.set_option("pair_kernels", 0)
    A.free()
    Bt.free()
