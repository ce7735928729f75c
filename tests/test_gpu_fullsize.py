"""BASELINE.json configurations 2-5 at their FULL per-GPU sizes, EVERY shard compared bit-exact with the
CPU oracle: the restated reference calls (oracle/roaring_oracle.c, bsi_oracle.c) run over all shards on
host threads from the same flattened descriptors the C ABI uploads (oracle/batch_oracle.c, pybatch.py).

  config 2   1024 shards x 2 dense rows        Bitmap.IntersectionCount roaring.go:711, Bitmap.Intersect :736
  config 3   256 shards x (64 mixed rows + F)  n-way Bitmap.Union :1272/:1410 then IntersectionCount; doTopK
                                               executor.go:2705; groupByIterator :8880 (decode in the matrix kernel)
  config 4   1024 shards x (32 x 32 + F) dense groupByIterator executor.go:8880-8934, per shard
  config 5   96 shards x 66 dense planes       fragment.rangeOp / rangeBetween fragment.go:937-1303, fragment.sum :724,
                                               fragment.min / max :754-853

Materialised results are downloaded through fbk_batch_download and decoded BY THE ORACLE before the
comparison, so no GPU kernel takes part in checking another.  The size-independent identities of the
earlier rounds are kept at the end as a second, independent check."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L
from oracle import pybatch as PB

pytestmark = pytest.mark.gpu


def _gpu_words(batch):
    """bit content of every row of a device batch, decoded by the oracle from the downloaded flat form"""
    d, p, n_rows = batch.download_flat()
    return PB.RowSet.from_flat(d, p, n_rows).words()


def test_config2_every_shard_vs_oracle(gpu_ctx):
    n = 1024
    wa, wb = D.dense_rows(n, 0.5, 2001), D.dense_rows(n, 0.5, 2002)
    A, B = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb)
    OA, OB = PB.RowSet.from_dense(wa), PB.RowSet.from_dense(wb)
    idx = np.arange(n)
    exp = PB.intersection_count(OA, idx, OB, idx)
    assert (gpu_ctx.intersection_count(A, idx, B, idx) == exp).all()
    plan = gpu_ctx.plan(A, idx, B, idx)
    plan.intersection_count_total()
    counts, total = plan.read(want_total=True)
    assert (counts == exp).all() and int(total) == int(exp.sum())
    plan.free()
    for op in (L.OP_AND, L.OP_XOR):
        out, cnt = gpu_ctx.setop(op, A, idx, B, idx)
        eo, ecnt = PB.setop(op, OA, idx, OB, idx)
        assert (cnt == ecnt).all()
        assert (_gpu_words(out) == eo.words()).all()
        out.free()
        eo.free()
    A.free()
    B.free()


@pytest.fixture(scope="module")
def config3():
    d, p, n_rows, groups, fd, fp, nbytes = D.config3_flat_subprocess(256, 64, 3000)
    return {"d": d, "p": p, "n_rows": n_rows, "groups": groups, "fd": fd, "fp": fp, "OA": PB.RowSet.from_flat(d, p, n_rows),
            "OF": PB.RowSet.from_flat(fd, fp, 256)}


def test_config3_union_of_64_every_shard_vs_oracle(gpu_ctx, config3):
    c = config3
    n_shards = c["groups"].shape[0]
    batch = gpu_ctx.upload_flat(c["d"], c["p"], c["n_rows"])
    F = gpu_ctx.upload_flat(c["fd"], c["fp"], n_shards)
    fidx = np.arange(n_shards)
    exp, exp_ucnt = PB.union_n_intersection_count(c["OA"], c["groups"], c["OF"], fidx)
    assert (gpu_ctx.union_n_intersection_count(batch, c["groups"], F, fidx) == exp).all()
    # the materialised union: cardinalities and bit content of all 256 x 16 result containers
    un, un_cnt = gpu_ctx.union_n(batch, c["groups"])
    assert (un_cnt == exp_ucnt).all()
    eu, _ = PB.union_n(c["OA"], c["groups"])
    assert (_gpu_words(un) == eu.words()).all()
    un.free()
    # ... and re-encoded by optimize(): same content
    un, un_cnt = gpu_ctx.union_n(batch, c["groups"], L.SETOP_OPTIMIZE)
    assert (un_cnt == exp_ucnt).all() and (_gpu_words(un) == eu.words()).all()
    un.free()
    batch.free()
    F.free()


def test_config3_topk_and_groupby_every_shard_vs_oracle(gpu_ctx, config3):
    c = config3
    g = c["groups"]
    n_shards = g.shape[0]
    batch = gpu_ctx.upload_flat(c["d"], c["p"], c["n_rows"])
    F = gpu_ctx.upload_flat(c["fd"], c["fp"], n_shards)
    fidx = np.arange(n_shards)
    # TopK / TopN shape: 64 rows x the filter row (k_rows_vs_filter)
    exp_t = PB.topk_counts(c["OA"], g, c["OF"], fidx)
    tot, ps = gpu_ctx.count_matrix(batch, g, F, fidx.reshape(-1, 1), per_shard=True)
    assert (ps[:, :, 0] == exp_t).all() and (tot[:, 0] == exp_t.sum(axis=0)).all()
    # GroupBy 32 x 32 + filter on mixed rows (k_count_matrix_fused), and without the filter
    exp_m = PB.count_matrix(c["OA"], g[:, :32], c["OA"], g[:, 32:], c["OF"], fidx)
    tot, ps = gpu_ctx.count_matrix(batch, g[:, :32], batch, g[:, 32:], F, fidx, per_shard=True)
    assert (ps == exp_m).all() and (tot == exp_m.sum(axis=0)).all()
    exp_nf = PB.count_matrix(c["OA"], g[:, :32], c["OA"], g[:, 32:])
    assert (gpu_ctx.count_matrix(batch, g[:, :32], batch, g[:, 32:], per_shard=True)[1] == exp_nf).all()
    batch.free()
    F.free()


def test_config3_row_pairs_every_pair_vs_oracle(gpu_ctx, config3):
    """RowSegment.IntersectionCount / Intersect / Union / Difference / Xor on config 3's mixed rows: row r
    against row r + 1 and against a row of very different density, in all 256 shards (k_icount, k_setop)."""
    c = config3
    g = c["groups"]
    batch = gpu_ctx.upload_flat(c["d"], c["p"], c["n_rows"])
    ra = np.concatenate([g[:, :-1].reshape(-1), g[:, :32].reshape(-1)])
    rb = np.concatenate([g[:, 1:].reshape(-1), g[:, :31:-1].reshape(-1)])
    assert (gpu_ctx.intersection_count(batch, ra, batch, rb) == PB.intersection_count(c["OA"], ra, c["OA"], rb)).all()
    sel = slice(0, None, 7)  # every 7th pair materialised (3400 pairs x 16 containers per operation)
    for op in (L.OP_AND, L.OP_OR, L.OP_XOR, L.OP_ANDNOT):
        for flags in (0, L.SETOP_OPTIMIZE):
            out, cnt = gpu_ctx.setop(op, batch, ra[sel], batch, rb[sel], flags)
            eo, ecnt = PB.setop(op, c["OA"], ra[sel], c["OA"], rb[sel])
            assert (cnt == ecnt).all(), (op, flags)
            assert (_gpu_words(out) == eo.words()).all(), (op, flags)
            out.free()
            eo.free()
    batch.free()


def test_config4_count_matrix_every_shard_vs_oracle(gpu_ctx):
    """The per-GPU slice of the 8192-shard configuration: 1024 shards x (32 x 32 rows + filter), 8.7 GB."""
    n_shards, n_a, n_b = 1024, 32, 32
    rng = D.rng_for(4100)
    wa = rng.integers(0, 2**64, (n_shards * n_a, 16, 1024), dtype=np.uint64)
    wb = rng.integers(0, 2**64, (n_shards * n_b, 16, 1024), dtype=np.uint64)
    wf = rng.integers(0, 2**64, (n_shards, 16, 1024), dtype=np.uint64)
    A, Bt, F = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb), gpu_ctx.upload_dense(wf)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)
    tot, ps = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf, per_shard=True)
    OA, OB, OF = PB.RowSet.from_dense(wa), PB.RowSet.from_dense(wb), PB.RowSet.from_dense(wf)
    del wa, wb
    exp = PB.count_matrix(OA, ra, OB, rb, OF, rf)
    assert (ps == exp).all() and (tot == exp.sum(axis=0)).all()
    # without the filter, on the first 128 shards
    exp_nf = PB.count_matrix(OA, ra[:128], OB, rb[:128])
    assert (gpu_ctx.count_matrix(A, ra[:128], Bt, rb[:128], per_shard=True)[1] == exp_nf).all()
    for b in (A, Bt, F):
        b.free()


def test_config4_as_surveyed_mixed_rows_every_shard_vs_oracle(gpu_ctx):
    """configs[3] on the input SURVEY.md 8d specifies: fields A and B of 32 rows each with densities LOG-UNIFORM in
    [0.001, 0.5] (two thirds of the rows are array rows; tests/datagen.py config4_flat), filter row p = 0.5, the 1024
    shards one GPU of eight owns.  Every per-shard 32 x 32 matrix against the restated groupByIterator
    (executor.go:8880-8934, oracle/batch_oracle.c), with and without the filter, one-shot call and prepared query, in every
    kernel the cost model can pick for encoded rows (fused / generic pairs)."""
    n_shards, n_a, n_b = 1024, 32, 32
    d, p, n_rows, groups, fd, fp, nbytes = D.config3_flat_subprocess(n_shards, n_a + n_b, 4000, config4=True)
    types = np.bincount(d["type"], minlength=4)
    assert types[3] == 0 and 0.55 < types[1] / (types[1] + types[2]) < 0.8, types  # ~2/3 arrays, no run containers
    OA, OF = PB.RowSet.from_flat(d, p, n_rows), PB.RowSet.from_flat(fd, fp, n_shards)
    batch, F = gpu_ctx.upload_flat(d, p, n_rows), gpu_ctx.upload_flat(fd, fp, n_shards)
    ga, gb, fidx = groups[:, :n_a], groups[:, n_a:], np.arange(n_shards)
    exp = PB.count_matrix(OA, ga, OA, gb, OF, fidx)
    exp_nf = PB.count_matrix(OA, ga, OA, gb)
    try:
        for fused in (-1, 1, 0):
            gpu_ctx.set_option("matrix_fused", fused)
            sl = slice(None) if fused != 0 else slice(0, 64)  # (the generic pair kernel: a slice is enough)
            tot, ps = gpu_ctx.count_matrix(batch, ga[sl], batch, gb[sl], F, fidx[sl], per_shard=True)
            assert (ps == exp[sl]).all() and (tot == exp[sl].sum(axis=0)).all(), fused
            assert (gpu_ctx.count_matrix(batch, ga[sl], batch, gb[sl], per_shard=True)[1] == exp_nf[sl]).all(), fused
        gpu_ctx.set_option("matrix_fused", -1)
        q = gpu_ctx.prepare_count_matrix(batch, ga, batch, gb, F, fidx, keep_per_shard=True)
        for _ in range(2):
            q.run()
            tot, ps = q.read(per_shard=True)
            assert (ps == exp).all() and (tot == exp.sum(axis=0)).all()
        q.free()
    finally:
        gpu_ctx.set_option("matrix_fused", -1)
    for b in (batch, F):
        b.free()
    for o in (OA, OF):
        o.free()


def test_config5_bsi_every_shard_vs_oracle(gpu_ctx):
    n_shards, depth = 96, 64
    w = D.dense_rows(n_shards * (depth + 2), 0.5, 5001).reshape(n_shards, depth + 2, 16, 1024)
    w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    w[-1, 0, 6:] = 0  # 100 000 000 columns: the last shard is partial
    w[-1, 0, 5, 900:] = 0  # 385 280 columns = 5 containers + 57 600 bits
    w[:, 1:] &= w[:, :1]  # planes only where a value exists (a well-formed BSI fragment)
    assert int(np.bitwise_count(w[:, 0]).sum()) == 100_000_000
    batch = gpu_ctx.upload_dense(w.reshape(-1))
    OA = PB.RowSet.from_dense(w.reshape(-1, 16, 1024))
    base = np.arange(n_shards, dtype=np.uint32) * (depth + 2)
    idx = np.arange(n_shards)
    for op, pred in ((L.BSI_GT, 1 << 62), (L.BSI_LTE, -(1 << 61) - 12345), (L.BSI_EQ, _some_value(w, 3)), (L.BSI_NEQ, 0)):
        got, gcnt = gpu_ctx.bsi_range(batch, base, op, depth, pred)
        exp, ecnt = PB.bsi_range(OA, base, depth, op, pred)
        assert (gcnt == ecnt).all(), (op, pred)
        assert (_gpu_words(got) == exp.words()).all(), (op, pred)
        # Sum(filter = the range result), two passes and the one-pass form
        es, ec = PB.bsi_sum(OA, base, depth, exp, idx)
        s, cn = gpu_ctx.bsi_sum(batch, base, depth, got, idx)
        assert (s == es).all() and (cn == ec).all(), (op, pred)
        s, cn = gpu_ctx.bsi_range_sum(batch, base, op, depth, pred)
        assert (s == es).all() and (cn == ec).all(), (op, pred)
        # Min / Max inside the range
        for is_max, fn in ((False, gpu_ctx.bsi_min), (True, gpu_ctx.bsi_max)):
            ev, evc = PB.bsi_minmax(OA, base, depth, is_max, exp, idx)
            v, vc = fn(batch, base, depth, got, idx)
            assert (v == ev).all() and (vc == evc).all(), (op, pred, is_max)
        got.free()
        exp.free()
    for lo, hi in ((-(1 << 62), 1 << 61), (1 << 40, (1 << 62) + 999), (-(1 << 63) + 1, -5)):
        got, gcnt = gpu_ctx.bsi_range_between(batch, base, depth, lo, hi)
        exp, ecnt = PB.bsi_range(OA, base, depth, PB.BETWEEN, lo, hi)
        assert (gcnt == ecnt).all() and (_gpu_words(got) == exp.words()).all(), (lo, hi)
        es, ec = PB.bsi_sum(OA, base, depth, exp, idx)
        s, cn = gpu_ctx.bsi_range_between_sum(batch, base, depth, lo, hi)
        assert (s == es).all() and (cn == ec).all(), (lo, hi)
        got.free()
        exp.free()
    es, ec = PB.bsi_sum(OA, base, depth)
    s, cn = gpu_ctx.bsi_sum(batch, base, depth)
    assert (s == es).all() and (cn == ec).all()
    batch.free()


def _some_value(w, shard):
    """the value of the first column of the shard that fits int64 (random planes: bit 63 is set in half of them)"""
    for col in range(1000):
        v = _value_of(w, shard, col)
        if -(1 << 63) < v < (1 << 63):
            return int(v)
    raise AssertionError("no column with a 63-bit value")


def _value_of(w, shard, col):
    """the stored value of one column of a BSI fragment given as words[shard][row][16][1024]"""
    sl, wd, bit = col >> 16, (col & 0xFFFF) >> 6, col & 63
    mag = sum(((int(w[shard, 2 + i, sl, wd]) >> bit) & 1) << i for i in range(w.shape[1] - 2))
    return -mag if (int(w[shard, 1, sl, wd]) >> bit) & 1 else mag


# ---- size-independent identities at the same sizes (kept from the earlier rounds: a second, independent check) ----


def test_config5_bsi_range_sum_properties_full_size(gpu_ctx):
    n_shards, depth = 96, 64
    w = D.dense_rows(n_shards * (depth + 2), 0.5, 5001).reshape(n_shards, depth + 2, 16, 1024)
    w[:, 0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    w[-1, 0, 6:] = 0
    w[:, 1:] &= w[:, :1]
    batch = gpu_ctx.upload_dense(w.reshape(-1))
    base = np.arange(n_shards, dtype=np.uint32) * (depth + 2)
    idx = np.arange(n_shards)
    exists = np.bitwise_count(w[:, 0]).sum(axis=(1, 2))
    k = 1 << 62
    gt, c_gt = gpu_ctx.bsi_range(batch, base, L.BSI_GT, depth, k)
    lte, c_lte = gpu_ctx.bsi_range(batch, base, L.BSI_LTE, depth, k)
    assert (c_gt + c_lte == exists.astype(np.uint64)).all()  # partition
    assert gpu_ctx.intersection_count(gt, idx, lte, idx).sum() == 0
    s_all, n_all = gpu_ctx.bsi_sum(batch, base, depth)
    s_gt, n_gt = gpu_ctx.bsi_sum(batch, base, depth, gt, idx)
    s_lte, n_lte = gpu_ctx.bsi_sum(batch, base, depth, lte, idx)
    assert (n_all == exists.astype(np.uint64)).all() and (n_gt == c_gt).all() and (n_lte == c_lte).all()
    assert ((s_gt.astype(np.uint64) + s_lte.astype(np.uint64)) == s_all.astype(np.uint64)).all()  # int64 wrap-around arithmetic
    for b in (gt, lte, batch):
        b.free()


def test_config4_count_matrix_properties(gpu_ctx):
    n_shards, n_a, n_b = 128, 32, 32
    wa = D.dense_rows(n_shards * n_a, 0.5, 4001)
    wb = D.dense_rows(n_shards * n_b, 0.5, 4002)
    wf = D.dense_rows(n_shards, 0.5, 4003)
    A, Bt, F = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb), gpu_ctx.upload_dense(wf)
    ra = np.arange(n_shards * n_a).reshape(n_shards, n_a)
    rb = np.arange(n_shards * n_b).reshape(n_shards, n_b)
    rf = np.arange(n_shards)
    tot = gpu_ctx.count_matrix(A, ra, Bt, rb, F, rf)
    assert (tot == gpu_ctx.count_matrix(Bt, rb, A, ra, F, rf).T).all()  # transpose symmetry
    wa3, wb3, wf3 = wa.reshape(n_shards, n_a, -1), wb.reshape(n_shards, n_b, -1), wf.reshape(n_shards, -1)
    for i, j in ((0, 0), (31, 31), (7, 19)):  # numpy: an oracle-independent model
        assert int(tot[i, j]) == int(np.bitwise_count(wa3[:, i] & wb3[:, j] & wf3).sum())
    assert (gpu_ctx.count_matrix(A, ra, Bt, rb) >= tot).all()
    for b in (A, Bt, F):
        b.free()
