"""The oracle's batch entry points (oracle/batch_oracle.c) add no algorithm of their own: they build
the oracle's Bitmaps from the flattened descriptors and run the restated reference calls per shard on
host threads.  Checked here against the per-object path through oracle/pyoracle.py / pybsi.py (which is
what the golden-vector tests pin) and against an independent numpy bitset model."""
import numpy as np

import datagen as D
from oracle import pybatch as PB
from oracle import pybsi as BS
from oracle import pyoracle as O


def _ocont(d, buf):
    p = buf[int(d["off"]):]
    if d["type"] == 1:
        return O.OContainer.array(p[: 2 * int(d["len"])].view(np.uint16))
    if d["type"] == 3:
        return O.OContainer.run(p[: 4 * int(d["len"])].view(np.uint16).reshape(-1, 2).tolist())
    return O.OContainer.bitmap(p[:8192].view(np.uint64), int(d["n"]))


def _obitmaps(flat):
    d, pay = flat.descs(), flat.payload()
    out = []
    for r in range(flat.n_rows):
        sel = d[d["row"] == r]
        out.append(O.OBitmap.from_containers([(int(e["key"]) & 15, _ocont(e, pay)) for e in sel]))
    return out


def _words_of_obitmap(b):
    w = np.zeros((16, 1024), dtype=np.uint64)
    for k, c in b.items():
        if c.p:
            w[k & 15] = c.words()
    return w


def test_flat_rowset_matches_per_object_path_on_config3_rows():
    n_shards, k = 3, 12
    rows, groups, filt = D.config3_flat(n_shards, k, seed_idx=3100, workers=1)
    A = PB.RowSet.from_flat(rows.descs(), rows.payload(), rows.n_rows)
    F = PB.RowSet.from_flat(filt.descs(), filt.payload(), filt.n_rows)
    bms, fbs = _obitmaps(rows), _obitmaps(filt)
    # bit content and counts of every row
    w = A.words(2)
    for r in range(rows.n_rows):
        assert (w[r] == _words_of_obitmap(bms[r])).all()
    assert A.counts().tolist() == [b.count() for b in bms]
    fidx = np.arange(n_shards)
    # Union-of-k then IntersectionCount
    got, ucnt = PB.union_n_intersection_count(A, groups, F, fidx, nthreads=3)
    for s in range(n_shards):
        u = bms[groups[s, 0]].union(*[bms[i] for i in groups[s, 1:]])
        assert int(ucnt[s]) == u.count() and int(got[s]) == u.intersection_count(fbs[s])
    U, ucnt2 = PB.union_n(A, groups)
    assert (ucnt2 == ucnt).all()
    uw = U.words(1)
    for s in range(n_shards):
        assert (uw[s] == _words_of_obitmap(bms[groups[s, 0]].union(*[bms[i] for i in groups[s, 1:]]))).all()
    # GroupBy matrix and TopK counts
    h = k // 2
    m = PB.count_matrix(A, groups[:, :h], A, groups[:, h:], F, fidx, nthreads=2)
    m_nf = PB.count_matrix(A, groups[:, :h], A, groups[:, h:])
    t = PB.topk_counts(A, groups, F, fidx)
    for s in range(n_shards):
        fa, fb = BS.Fragment([bms[i] for i in groups[s, :h]]), BS.Fragment([bms[i] for i in groups[s, h:]])
        assert (m[s] == BS.groupby_counts(fa, fb, fbs[s])).all()
        assert (m_nf[s] == BS.groupby_counts(fa, fb, None)).all()
        assert t[s].tolist() == [bms[i].intersection_count(fbs[s]) for i in groups[s]]
    # pair ops: every row against its neighbour
    ra, rb = np.arange(rows.n_rows - 1), np.arange(1, rows.n_rows)
    ic = PB.intersection_count(A, ra, A, rb)
    assert ic.tolist() == [bms[a].intersection_count(bms[b]) for a, b in zip(ra, rb)]
    for op, fn in ((PB.OP_AND, lambda a, b: a.intersect(b)), (PB.OP_OR, lambda a, b: a.union(b)), (PB.OP_XOR, lambda a, b: a.xor(b)),
                   (PB.OP_ANDNOT, lambda a, b: a.difference(b))):
        R, cnt = PB.setop(op, A, ra[:6], A, rb[:6])
        rw = R.words()
        for i in range(6):
            e = fn(bms[ra[i]], bms[rb[i]])
            assert int(cnt[i]) == e.count() and (rw[i] == _words_of_obitmap(e)).all()


def test_dense_rowset_against_numpy():
    n_shards, na, nb = 4, 3, 5
    wa = D.dense_rows(n_shards * na, 0.5, 7101)
    wb = D.dense_rows(n_shards * nb, 0.25, 7102)
    wf = D.dense_rows(n_shards, 0.5, 7103)
    wf[1, 3] = 0  # a nil container in a filter row
    A, B, F = PB.RowSet.from_dense(wa), PB.RowSet.from_dense(wb), PB.RowSet.from_dense(wf)
    assert (A.words() == wa).all() and (F.words() == wf).all()
    ra, rb = np.arange(n_shards * na).reshape(n_shards, na), np.arange(n_shards * nb).reshape(n_shards, nb)
    m = PB.count_matrix(A, ra, B, rb, F, np.arange(n_shards))
    for s in range(n_shards):
        for i in range(na):
            for j in range(nb):
                assert int(m[s, i, j]) == int(np.bitwise_count(wa[ra[s, i]] & wb[rb[s, j]] & wf[s]).sum())
    ic1 = PB.intersection_count(A, ra[:, 0], B, rb[:, 0], nthreads=1)
    ic4 = PB.intersection_count(A, ra[:, 0], B, rb[:, 0], nthreads=4)
    assert (ic1 == ic4).all()
    assert ic1.tolist() == [int(np.bitwise_count(wa[a] & wb[b]).sum()) for a, b in zip(ra[:, 0], rb[:, 0])]


def test_batch_bsi_against_per_shard_calls_and_numpy():
    n_shards, depth = 3, 12
    rng = D.rng_for(7201)
    w = rng.integers(0, 2**64, (n_shards, depth + 2, 16, 1024), dtype=np.uint64)
    w[:, 0] |= rng.integers(0, 2**64, (n_shards, 16, 1024), dtype=np.uint64)  # 75 % exist
    w[-1, 0, 9:] = 0
    w[:, 1:] &= w[:, :1]
    A = PB.RowSet.from_dense(w.reshape(-1, 16, 1024))
    base = np.arange(n_shards) * (depth + 2)
    frs = []
    for s in range(n_shards):
        frs.append(BS.Fragment([O.OBitmap.from_containers([(sl, O.OContainer.bitmap(w[s, r, sl])) for sl in range(16) if w[s, r, sl].any()])
                                for r in range(depth + 2)]))
    for op, pred in ((PB.GT, 1000), (PB.LTE, -7), (PB.EQ, 5), (PB.NEQ, 0), (PB.LT, 1 << 11), (PB.GTE, -(1 << 11) + 1)):
        R, cnt = PB.bsi_range(A, base, depth, op, pred, nthreads=2)
        rw = R.words()
        for s in range(n_shards):
            e = BS.bsi_range(frs[s], op, depth, pred)
            assert int(cnt[s]) == e.count() and (rw[s] == _words_of_obitmap(e)).all()
        ssum, scnt = PB.bsi_sum(A, base, depth, R, np.arange(n_shards))
        for s in range(n_shards):
            assert (int(ssum[s]), int(scnt[s])) == BS.bsi_sum(frs[s], BS.bsi_range(frs[s], op, depth, pred), True)
    R, cnt = PB.bsi_range(A, base, depth, PB.BETWEEN, -300, 2000)
    for s in range(n_shards):
        assert int(cnt[s]) == BS.bsi_range_between(frs[s], depth, -300, 2000).count()
    # Sum without a filter against the definition on numpy popcounts
    ssum, scnt = PB.bsi_sum(A, base, depth)
    for s in range(n_shards):
        pos, neg = w[s, 0] & ~w[s, 1], w[s, 0] & w[s, 1]
        tot = sum((int(np.bitwise_count(pos & w[s, 2 + i]).sum()) - int(np.bitwise_count(neg & w[s, 2 + i]).sum())) << i for i in range(depth))
        assert int(ssum[s]) == tot and int(scnt[s]) == int(np.bitwise_count(w[s, 0]).sum())
    for is_max in (False, True):
        v, c = PB.bsi_minmax(A, base, depth, is_max)
        for s in range(n_shards):
            assert (int(v[s]), int(c[s])) == (BS.bsi_max if is_max else BS.bsi_min)(frs[s], None, depth)


def test_malformed_descriptor_table_is_refused():
    rows, _, _ = D.config3_flat(1, 2, seed_idx=3101, workers=1)
    d = rows.descs().copy()
    d["off"][-1] = rows.bytes  # payload out of bounds
    try:
        PB.RowSet.from_flat(d, rows.payload(), rows.n_rows)
        assert False, "accepted an out-of-bounds payload offset"
    except ValueError:
        pass
