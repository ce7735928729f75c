"""Prepared queries (fbk_query_*): the launch-only forms of the count matrix, the n-way fold with its fused count
and BSI Sum / one-pass Sum(Range).  Every result is compared with the CPU oracle (oracle/batch_oracle.c over the
same flattened rows: groupByIterator executor.go:8880, Bitmap.Union roaring.go:1272 + IntersectionCount :711,
fragment.sum fragment.go:724, rangeOp :937) and with the one-shot call; repeated runs, caller-owned device
buffers and the accumulate flag included."""
import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L
from oracle import pybatch as PB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mixed():
    rows, groups, filt = D.config3_flat(12, 16, seed_idx=8100, workers=1)
    return rows, groups, filt, PB.RowSet.from_flat(rows.descs(), rows.payload(), rows.n_rows), PB.RowSet.from_flat(filt.descs(), filt.payload(), filt.n_rows)


def test_prepared_count_matrix_mixed_and_dense(gpu_ctx, mixed):
    import torch

    rows, g, filt, OA, OF = mixed
    n = g.shape[0]
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    F = gpu_ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
    fidx = np.arange(n)
    for ra, rb, f in ((g[:, :8], g[:, 8:], True), (g[:, :12], g[:, 12:], False), (g, fidx.reshape(-1, 1), None)):
        if f is None:  # TopK shape: rows x the filter row as B
            exp = PB.topk_counts(OA, ra, OF, fidx)[:, :, None]
            q = gpu_ctx.prepare_count_matrix(batch, ra, F, rb, keep_per_shard=True)
        else:
            exp = PB.count_matrix(OA, ra, OA, rb, OF if f else None, fidx if f else None)
            q = gpu_ctx.prepare_count_matrix(batch, ra, batch, rb, F if f else None, fidx if f else None, keep_per_shard=True)
        for _ in range(3):  # every run overwrites
            q.run()
        tot, ps = q.read(per_shard=True)
        assert (ps == exp).all() and (tot == exp.sum(axis=0)).all()
        # a caller-owned device buffer, accumulated into twice on top of a known value
        cell = torch.full((exp.shape[1] * exp.shape[2],), 5, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        q.run(cell.data_ptr(), accumulate=True)
        q.run(cell.data_ptr(), accumulate=True)
        gpu_ctx.synchronize()
        assert (cell.cpu().numpy().view(np.uint64).reshape(exp.shape[1:]) == 2 * exp.sum(axis=0) + 5).all()
        q.run(cell.data_ptr())  # without the flag: overwritten
        gpu_ctx.synchronize()
        assert (cell.cpu().numpy().view(np.uint64).reshape(exp.shape[1:]) == exp.sum(axis=0)).all()
        q.free()
    # dense rows: the matrix-core kernel
    wa, wb, wf = D.dense_rows(4 * 33, 0.5, 8201), D.dense_rows(4 * 40, 0.5, 8202), D.dense_rows(4, 0.5, 8203)
    A, B, Ff = gpu_ctx.upload_dense(wa), gpu_ctx.upload_dense(wb), gpu_ctx.upload_dense(wf)
    ra, rb = np.arange(4 * 33).reshape(4, 33), np.arange(4 * 40).reshape(4, 40)
    exp = PB.count_matrix(PB.RowSet.from_dense(wa), ra, PB.RowSet.from_dense(wb), rb, PB.RowSet.from_dense(wf), np.arange(4))
    q = gpu_ctx.prepare_count_matrix(A, ra, B, rb, Ff, np.arange(4))
    q.run()
    assert (q.read() == exp.sum(axis=0)).all()
    assert (gpu_ctx.count_matrix(A, ra, B, rb, Ff, np.arange(4)) == exp.sum(axis=0)).all()
    q.free()
    for b in (A, B, Ff, batch, F):
        b.free()


def test_prepared_fold_intersection_count(gpu_ctx, mixed):
    rows, g, filt, OA, OF = mixed
    n = g.shape[0]
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    F = gpu_ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
    fidx = np.arange(n)
    exp, exp_u = PB.union_n_intersection_count(OA, g, OF, fidx)
    q = gpu_ctx.prepare_fold_intersection_count(L.OP_OR, batch, g, F, fidx)
    q.run()
    q.run()
    assert (q.read() == exp).all()
    q.run(accumulate=True)
    assert (q.read() == 2 * exp).all()
    q.free()
    q = gpu_ctx.prepare_fold_intersection_count(L.OP_OR, batch, g)  # no filter: |∪ rows|
    q.run()
    assert (q.read() == exp_u).all()
    q.free()
    # Intersect of the four densest rows of each shard
    gi = g[:, :4]
    R, _ = PB.setop(PB.OP_AND, OA, gi[:, 0], OA, gi[:, 1])
    for k in (2, 3):
        R, cnt = PB.setop(PB.OP_AND, R, np.arange(n), OA, gi[:, k])
    q = gpu_ctx.prepare_fold_intersection_count(L.OP_AND, batch, gi)
    q.run()
    assert (q.read() == cnt).all()
    q.free()
    batch.free()
    F.free()


def test_prepared_bsi_sum_and_one_pass_range_sum(gpu_ctx):
    n_shards, depth = 7, 20
    rng = D.rng_for(8301)
    w = rng.integers(0, 2**64, (n_shards, depth + 2, 16, 1024), dtype=np.uint64)
    w[:, 0] |= rng.integers(0, 2**64, (n_shards, 16, 1024), dtype=np.uint64)
    w[-1, 0, 11:] = 0
    w[:, 1:] &= w[:, :1]
    wf = rng.integers(0, 2**64, (n_shards, 16, 1024), dtype=np.uint64)
    batch, F = gpu_ctx.upload_dense(w.reshape(-1)), gpu_ctx.upload_dense(wf)
    OA, OF = PB.RowSet.from_dense(w.reshape(-1, 16, 1024)), PB.RowSet.from_dense(wf)
    base, idx = np.arange(n_shards) * (depth + 2), np.arange(n_shards)
    for filt in (False, True):
        es, ec = PB.bsi_sum(OA, base, depth, OF if filt else None, idx if filt else None)
        q = gpu_ctx.prepare_bsi_sum(batch, base, depth, filt=F if filt else None, rows_f=idx if filt else None)
        q.run()
        q.run()
        s, c = q.read()
        assert (s == es).all() and (c == ec).all()
        q.free()
    for op, pred in ((L.BSI_GT, 1234), (L.BSI_LTE, -777), (L.BSI_LT, 1 << 19), (L.BSI_GTE, -(1 << 18))):
        R, _ = PB.bsi_range(OA, base, depth, op, pred)
        for filt in (False, True):
            if filt:
                RF, _ = PB.setop(PB.OP_AND, R, idx, OF, idx)
            es, ec = PB.bsi_sum(OA, base, depth, RF if filt else R, idx)
            q = gpu_ctx.prepare_bsi_sum(batch, base, depth, op, pred, F if filt else None, idx if filt else None)
            q.run()
            s, c = q.read()
            assert (s == es).all() and (c == ec).all(), (op, pred, filt)
            q.free()
    # a predicate only the two-pass form serves is refused at prepare time, with a message on the context
    with pytest.raises(L.FbkError):
        gpu_ctx.prepare_bsi_sum(batch, base, depth, L.BSI_EQ, 5)
    assert "one-pass" in gpu_ctx.last_error()[1]
    batch.free()
    F.free()


def test_prepared_query_argument_errors(gpu_ctx, mixed):
    rows, g, filt, OA, OF = mixed
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    with pytest.raises(L.FbkError):
        gpu_ctx.prepare_count_matrix(batch, g[:, :4] + 100000, batch, g[:, 4:8])  # row index out of range
    q = gpu_ctx.prepare_count_matrix(batch, g[:, :4], batch, g[:, 4:8])
    with pytest.raises(L.FbkError):
        q.read()  # before the first run
    other = gpu_ctx.fork()
    with pytest.raises(L.FbkError):
        L.check(other.lib.fbk_query_run(other.h, q.h, None, 0))  # prepared on another context
    other.close()
    q.free()
    batch.free()


def _words_of(batch):
    """bit content of every row of a batch: [n_rows, 16, 1024] uint64"""
    rows = batch.download()
    w = np.zeros((len(rows), 16, 1024), dtype=np.uint64)
    for r, row in enumerate(rows):
        for k, c in row.items():
            w[r, k & 15] = c.words()
    return w


def test_prepared_bsi_range_rows_stay_on_the_device(gpu_ctx):
    """fbk_query_bsi_range: Row(v op k) as a launch-only query whose result rows feed the next operator without leaving the
    device — Range(> k) then Sum(filter = that row), BASELINE config 5's pipeline, both prepared."""
    n_shards, depth = 6, 20
    rng = D.rng_for(8401)
    w = rng.integers(0, 2**64, (n_shards, depth + 2, 16, 1024), dtype=np.uint64)
    w[:, 0] |= rng.integers(0, 2**64, (n_shards, 16, 1024), dtype=np.uint64)
    w[-1, 0, 9:] = 0
    w[:, 1:] &= w[:, :1]
    batch = gpu_ctx.upload_dense(w.reshape(-1))
    OA = PB.RowSet.from_dense(w.reshape(-1, 16, 1024))
    base, idx = np.arange(n_shards) * (depth + 2), np.arange(n_shards)
    for op, pred in ((L.BSI_GT, 4321), (L.BSI_LTE, -99), (L.BSI_EQ, 77), (L.BSI_NEQ, 0), (L.BSI_LT, -(1 << 30)), (L.BSI_GTE, 0)):
        R, ecnt = PB.bsi_range(OA, base, depth, op, pred)
        q = gpu_ctx.prepare_bsi_range(batch, base, op, depth, pred)
        with pytest.raises(L.FbkError):
            q.output()  # no run yet
        for _ in range(2):  # every run rewrites the same output batch
            q.run()
        assert (q.read() == ecnt).all(), (op, pred)
        out = q.output()
        assert (_words_of(out) == R.words()).all(), (op, pred)
        # the borrowed rows as the filter of a prepared Sum and of the one-shot call
        es, ec = PB.bsi_sum(OA, base, depth, R, idx)
        s, c = gpu_ctx.bsi_sum(batch, base, depth, out, idx)
        assert (s == es).all() and (c == ec).all(), (op, pred)
        qs = gpu_ctx.prepare_bsi_sum(batch, base, depth, filt=out, rows_f=idx)
        q.run()
        qs.run()
        s, c = qs.read()
        assert (s == es).all() and (c == ec).all(), (op, pred)
        # the one-shot call agrees
        o1, c1 = gpu_ctx.bsi_range(batch, base, op, depth, pred)
        assert (c1 == ecnt).all() and (_words_of(o1) == R.words()).all()
        o1.free()
        qs.free()
        q.free()
    with pytest.raises(L.FbkError):
        gpu_ctx.prepare_bsi_range(batch, base, 9, depth, 1)  # ErrInvalidRangeOperation
    with pytest.raises(L.FbkError):
        gpu_ctx.prepare_bsi_range(batch, base + 10**6, L.BSI_GT, depth, 1)
    batch.free()


@pytest.mark.parametrize("flags", [0, L.SETOP_OPTIMIZE])
def test_prepared_fold_materialised(gpu_ctx, mixed, flags):
    """fbk_query_fold: Union / Xor / Difference of k rows per group, the result rows kept on the device; with FBK_SETOP_OPTIMIZE
    the descriptors and payload bytes equal the one-shot fbk_fold_n's."""
    rows, g, filt, OA, OF = mixed
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    eu, ucnt = PB.union_n(OA, g)
    q = gpu_ctx.prepare_fold(L.OP_OR, batch, g, flags)
    for _ in range(2):
        q.run()
    assert (q.read() == ucnt).all()
    out = q.output()
    assert (_words_of(out) == eu.words()).all()
    for op in (L.OP_OR, L.OP_XOR, L.OP_ANDNOT, L.OP_AND):
        qq = gpu_ctx.prepare_fold(op, batch, g[:, :5], flags)
        qq.run()
        o1, c1 = gpu_ctx.fold_n(op, batch, g[:, :5], flags)
        assert (qq.read() == c1).all()
        d0, p0, _ = qq.output().download_flat()
        d1, p1, _ = o1.download_flat()
        assert d0.tobytes() == d1.tobytes() and p0.tobytes() == p1.tobytes(), op
        assert qq.output().to_roaring() == o1.to_roaring()
        o1.free()
        qq.free()
    q.free()
    with pytest.raises(L.FbkError):
        gpu_ctx.prepare_fold(L.OP_AND, batch, g[:, :0], flags)  # Intersect needs at least one row per group (executor.go:5363)
    batch.free()


def test_prepared_topn_orders_on_the_device(gpu_ctx, mixed):
    """fbk_query_topn against the one-shot fbk_topn (both ordering paths of which are tied to the oracle in test_gpu_topn.py)
    and the oracle's per-shard counts: thresholds, n, with and without a source row, repeated runs."""
    rows, g, filt, OA, OF = mixed
    n = g.shape[0]
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    F = gpu_ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
    fidx = np.arange(n)
    exp_counts = PB.topk_counts(OA, g, OF, fidx).sum(axis=0)
    for use_f in (True, False):
        for top, mt, tt in ((0, 0, 0), (5, 0, 0), (1, 0, 0), (0, 3000, 0), (4, 0, 30), (0, 0, 90)):
            if tt and not use_f:
                continue
            fa = (F, fidx) if use_f else (None, None)
            e_idx, e_cnt = gpu_ctx.topn(batch, g, top, *fa, min_threshold=mt, tanimoto_threshold=tt)
            q = gpu_ctx.prepare_topn(batch, g, top, *fa, min_threshold=mt, tanimoto_threshold=tt)
            for _ in range(2):
                q.run()
            idx, cnt = q.read()
            assert idx.tolist() == e_idx.tolist() and cnt.tolist() == e_cnt.tolist(), (use_f, top, mt, tt)
            if use_f and not mt and not tt:
                assert all(int(c) == int(exp_counts[i]) for i, c in zip(idx, cnt))
                if top == 0:
                    assert len(idx) == int((exp_counts != 0).sum())
            q.free()
    with pytest.raises(L.FbkError):
        gpu_ctx.prepare_topn(batch, g, 3, F, fidx, tanimoto_threshold=101)
    batch.free()
    F.free()


def test_prepared_row_records_follow_a_rewritten_batch(gpu_ctx, mixed):
    """Prepared folds / TopN resolve the descriptors of every (group / shard, slot) into contiguous records on their first run
    (k_resolve_rows).  Same results as the one-shot calls, which do not use them; and when the batch the query reads is REWRITTEN
    on the device (here: the output of a plan, re-run with another operation) the next run resolves again instead of chasing the
    old descriptors."""
    rows, g, filt, OA, OF = mixed
    n = g.shape[0]
    batch = gpu_ctx.upload_flat(rows.descs(), rows.payload(), rows.n_rows)
    F = gpu_ctx.upload_flat(filt.descs(), filt.payload(), filt.n_rows)
    fidx = np.arange(n)
    try:
        # prepared (resolved records) against the one-shot calls, whose kernels gather the descriptors through the row lists
        q1 = gpu_ctx.prepare_fold_intersection_count(L.OP_OR, batch, g, F, fidx)
        q2 = gpu_ctx.prepare_fold(L.OP_XOR, batch, g[:, :7], L.SETOP_OPTIMIZE)
        q3 = gpu_ctx.prepare_topn(batch, g, 0, F, fidx)
        q4 = gpu_ctx.prepare_count_matrix(batch, g, F, fidx.reshape(-1, 1))
        for q in (q1, q2, q3, q4):
            q.run()
            q.run()
        d2, p2, _ = q2.output().download_flat()
        o2, c2 = gpu_ctx.fold_n(L.OP_XOR, batch, g[:, :7], L.SETOP_OPTIMIZE)
        e_d2, e_p2, _ = o2.download_flat()
        assert q1.read().tolist() == gpu_ctx.fold_n_intersection_count(L.OP_OR, batch, g, F, fidx).tolist()
        assert q1.read().tolist() == PB.union_n_intersection_count(OA, g, OF, fidx)[0].tolist()
        assert q2.read().tolist() == c2.tolist() and d2.tobytes() == e_d2.tobytes() and p2.tobytes() == e_p2.tobytes()
        e_idx, e_cnt = gpu_ctx.topn(batch, g, 0, F, fidx)
        idx, cnt = q3.read()
        assert idx.tolist() == e_idx.tolist() and cnt.tolist() == e_cnt.tolist()
        assert (q4.read() == gpu_ctx.count_matrix(batch, g, F, fidx.reshape(-1, 1))).all()
        o2.free()
        for q in (q1, q2, q3, q4):
            q.free()
        # a query over a batch that changes under it
        ia, ib = g[:, :6].reshape(-1), g[:, 6:12].reshape(-1)  # 6 output rows per shard
        plan = gpu_ctx.plan(batch, ia, batch, ib)
        plan.setop(L.OP_OR)
        O = plan.output()
        og = np.arange(n * 6).reshape(n, 6)
        q = gpu_ctx.prepare_fold_intersection_count(L.OP_OR, O, og, F, fidx)
        qt = gpu_ctx.prepare_topn(O, og, 0, F, fidx)
        qm = gpu_ctx.prepare_count_matrix(O, og, O, og[:, ::-1].copy(), F, fidx)  # 6 x 6 rows: the kernel with the prepared program
        for op in (L.OP_OR, L.OP_AND, L.OP_XOR):
            plan.setop(op)  # rewrites O in place
            q.run()
            qt.run()
            # the count matrix's program (row tables, resolved array items: addresses INTO O's arena) follows the rewrite too
            qm.run()
            qm.run()
            gpu_ctx.set_option("matrix_fused", 0)
            try:
                e_m = gpu_ctx.count_matrix(O, og, O, og[:, ::-1].copy(), F, fidx)
            finally:
                gpu_ctx.set_option("matrix_fused", -1)
            assert (qm.read() == e_m).all(), op
            assert q.read().tolist() == gpu_ctx.fold_n_intersection_count(L.OP_OR, O, og, F, fidx).tolist(), op
            e_idx, e_cnt = gpu_ctx.topn(O, og, 0, F, fidx)
            idx, cnt = qt.read()
            assert idx.tolist() == e_idx.tolist() and cnt.tolist() == e_cnt.tolist(), op
        q.free()
        qt.free()
        qm.free()
        plan.free()
    finally:
        gpu_ctx.set_option("matrix_fused", -1)
    batch.free()
    F.free()
