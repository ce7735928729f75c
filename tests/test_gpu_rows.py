"""fbk_rows (fragment.rows / executeRowsShard) against the oracle's restatement of the reference's filter
protocol (oracle/pyfilter.py <- roaring/filter.go): BitmapRowFilter over BitmapColumnFilter and
BitmapRowLimitFilter, driven container by container in key order by ApplyFilterToIterator."""
import numpy as np
import pytest

import datagen as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def PF():
    from oracle import pyfilter

    return pyfilter


def upload_fragment(ctx, frag):
    """frag: row id -> {slot: oracle container}; one device row per row id, ascending."""
    ids = sorted(frag)
    batch = ctx.upload([{s: D.to_fbk(c) for s, c in frag[r].items()} for r in ids])
    return batch, ids


def oracle_rows(PF, frag, start=0, column=None, limit=0, stats=None):
    cont = {r * 16 + s: PF.wrap(c) for r, row in frag.items() for s, c in row.items()}
    filters = ([PF.ColumnFilter(column)] if column is not None else []) + ([PF.RowLimitFilter(limit)] if limit else [])
    return PF.fragment_rows(cont, start, filters, stats)


def gpu_rows(ctx, batch, ids, start=0, column=None, limit=0):
    first = int(np.searchsorted(ids, start))
    cand = np.arange(first, len(ids), dtype=np.uint32)  # the fragment's rows from `start` on = device rows first ..
    pos = ctx.rows(batch, cand, column, limit, row_ids=np.asarray(ids[first:], dtype=np.uint64))
    return [ids[first + int(p)] for p in pos]


def test_reference_sample_data(gpu_ctx, oracle, PF):
    """filter_internal_test.go's sample fragment (requireSampleData :24-40) and the expectations of TestBaseFilter /
    TestColumnFilter, through the C ABI."""
    O = oracle
    frag = {}
    for i in range(1, 16):
        for row in range(0, 100, i):
            frag.setdefault(row, {})[i] = O.OContainer.array([i])
    batch, ids = upload_fragment(gpu_ctx, frag)
    assert gpu_rows(gpu_ctx, batch, ids) == list(range(100))
    for i in range(1, 16):
        assert gpu_rows(gpu_ctx, batch, ids, column=(i << 16) + i) == list(range(0, 100, i))
        assert gpu_rows(gpu_ctx, batch, ids, column=(i << 16) + i + 1) == []
    # TestRowsFilter's "limit" case without its rows filter: the first matching row of stride 2
    assert gpu_rows(gpu_ctx, batch, ids, column=(2 << 16) + 2, limit=1) == oracle_rows(PF, frag, 0, (2 << 16) + 2, 1) == [0]
    batch.free()


def test_rows_iteration_vectors(gpu_ctx, oracle):
    """TestFragment_RowsIteration (fragment_internal_test.go:3016-3110): firstContainer, secondRow."""
    O = oracle
    frag = {i: {0: O.OContainer.array([i % 2])} for i in range(100, 200)}
    batch, ids = upload_fragment(gpu_ctx, frag)
    assert gpu_rows(gpu_ctx, batch, ids) == list(range(100, 200))
    assert gpu_rows(gpu_ctx, batch, ids, column=1) == [i for i in range(100, 200) if i % 2]
    batch.free()
    frag = {1: {1: O.OContainer.array([66000 & 0xFFFF])}, 2: {1: O.OContainer.array([66000 & 0xFFFF]), 2: O.OContainer.array([166000 & 0xFFFF])}}
    batch, ids = upload_fragment(gpu_ctx, frag)
    assert gpu_rows(gpu_ctx, batch, ids) == [1, 2]
    assert gpu_rows(gpu_ctx, batch, ids, column=66000) == [1, 2]
    assert gpu_rows(gpu_ctx, batch, ids, column=166000) == [2]
    batch.free()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_fragment_every_encoding(gpu_ctx, oracle, PF, seed):
    """Random fragment: rows with containers of every encoding (and rows without any), columns probed inside
    arrays, runs and bitmaps, start rows, limits — the limit rule is the reference's (the limit filter counts
    the non-empty rows it is asked about, matching or not)."""
    O = oracle
    rng = D.rng_for(900 + seed)
    frag = {}
    for r in sorted(rng.choice(3000, size=400, replace=False)):
        row = D.random_row(rng, 0, p_missing=0.6)
        frag[int(r)] = {k & 15: c for k, c in row.items() if c.n}  # (the reference never stores an empty container)
        if not frag[int(r)]:
            del frag[int(r)]
    batch, ids = upload_fragment(gpu_ctx, frag)
    assert gpu_rows(gpu_ctx, batch, ids) == oracle_rows(PF, frag) == ids
    cols = []
    for _ in range(12):  # columns that some row really holds, and their neighbours
        r = ids[int(rng.integers(0, len(ids)))]
        s = sorted(frag[r])[int(rng.integers(0, len(frag[r])))]
        w = frag[r][s].words()
        bits = np.flatnonzero(np.unpackbits(w.view(np.uint8), bitorder="little"))
        v = int(bits[int(rng.integers(0, bits.size))])
        cols += [(s << 16) + v, (s << 16) + ((v + 1) & 0xFFFF)]
    for col in cols:
        for start in (0, ids[len(ids) // 3]):
            for limit in (0, 1, 7, 10000):
                got = gpu_rows(gpu_ctx, batch, ids, start, col, limit)
                assert got == oracle_rows(PF, frag, start, col, limit), (col, start, limit)
    for limit in (1, 5, 399, 400, 401):
        assert gpu_rows(gpu_ctx, batch, ids, 0, None, limit) == oracle_rows(PF, frag, 0, None, limit) == ids[:limit]
    batch.free()


def test_limit_with_column_on_adjacent_rows(gpu_ctx, oracle, PF):
    """Dense row ids: the column filter's skip to (next row, column's slot) hides a directly following row whose
    containers all lie in lower slots, so the limit filter does not count it (found by re-seeding the random
    fragment test, scripts/fuzz_parity.sh).  Hand-made cases, then random dense fragments."""
    O = oracle
    one = lambda *v: O.OContainer.array(list(v))
    col = (5 << 16) + 9
    # row 11 has only slot 2 (< 5) and follows row 10 (slot 5 seen): not counted -> row 12 is still within limit 2
    frag = {10: {5: one(9)}, 11: {2: one(1)}, 12: {5: one(9)}, 13: {7: one(3)}, 14: {5: one(9)}}
    batch, ids = upload_fragment(gpu_ctx, frag)
    for limit in (1, 2, 3, 4):
        got = gpu_rows(gpu_ctx, batch, ids, 0, col, limit)
        assert got == oracle_rows(PF, frag, 0, col, limit), limit
    assert gpu_rows(gpu_ctx, batch, ids, 0, col, 2) == [10, 12]
    # the same rows two ids apart: row 22 is looked at and counted -> limit 2 ends before row 24
    far = {2 * r: v for r, v in frag.items()}
    b2, ids2 = upload_fragment(gpu_ctx, far)
    assert gpu_rows(gpu_ctx, b2, ids2, 0, col, 2) == oracle_rows(PF, far, 0, col, 2) == [20]
    # starting AT the hidden row: nothing precedes it in this scan, it is counted
    assert gpu_rows(gpu_ctx, batch, ids, 11, col, 1) == oracle_rows(PF, frag, 11, col, 1) == []
    assert gpu_rows(gpu_ctx, batch, ids, 11, col, 2) == oracle_rows(PF, frag, 11, col, 2) == [12]
    batch.free()
    b2.free()
    rng = D.rng_for(907)
    for _ in range(6):
        frag = {}
        for r in range(int(rng.integers(0, 5)), 260):
            if rng.random() < 0.8:
                frag[r] = {int(s): one(*sorted({int(v) for v in rng.integers(0, 4, size=2)})) for s in rng.choice(16, size=int(rng.integers(1, 4)), replace=False)}
        batch, ids = upload_fragment(gpu_ctx, frag)
        for _ in range(8):
            col = (int(rng.integers(0, 16)) << 16) + int(rng.integers(0, 4))
            start = int(rng.integers(0, 200))
            for limit in (1, 3, 10, 60):
                assert gpu_rows(gpu_ctx, batch, ids, start, col, limit) == oracle_rows(PF, frag, start, col, limit), (col, start, limit)
        batch.free()
    # a column together with a limit needs the ids
    batch, ids = upload_fragment(gpu_ctx, {1: {0: one(1)}})
    with pytest.raises(Exception, match="row_ids"):
        gpu_ctx.rows(batch, [0], column=1, limit=1)
    with pytest.raises(Exception, match="ascending"):
        gpu_ctx.rows(batch, [0, 0], column=1, limit=1, row_ids=[5, 5])
    batch.free()


def test_whole_fragment_scan_many_rows(gpu_ctx, oracle, PF):
    """A fragment of 200 000 sparse rows (a field with that many row ids, one or two small containers per row) in
    ONE call: the device looks at every row at once where the protocol walks the keys in order and skips ahead
    — same answer, and the protocol's skip-ahead statistics show what it opens (one container per row)."""
    O = oracle
    rng = D.rng_for(77)
    n = 200_000
    slots = rng.integers(0, 16, n)
    vals = rng.integers(0, 65536, n)
    col = (5 << 16) + 4242
    hit = rng.random(n) < 0.01
    conts = []
    for i in range(n):
        if hit[i]:
            conts.append({5: O.OContainer.array(sorted({4242, int(vals[i])}))})
        else:
            conts.append({int(slots[i]): O.OContainer.array([int(vals[i])])})
    batch = gpu_ctx.upload([{s: D.to_fbk(c) for s, c in row.items()} for row in conts])
    got = gpu_ctx.rows(batch, np.arange(n, dtype=np.uint32), col)
    exp = [i for i in range(n) if 5 in conts[i] and (hit[i] or (slots[i] == 5 and vals[i] == 4242))]
    assert got.tolist() == exp
    # the protocol on a 20 000-row prefix (pure Python): identical rows, and it opened at most one container per row
    sub = {i: conts[i] for i in range(20_000)}
    stats = {}
    assert oracle_rows(PF, sub, 0, col, 0, stats) == [i for i in exp if i < 20_000]
    assert stats["consider_data"] <= 20_000
    assert gpu_ctx.rows(batch, np.arange(n, dtype=np.uint32), None, 50).tolist() == list(range(50))
    batch.free()


def test_rows_argument_errors(gpu_ctx, oracle):
    from featurebase_amd.lib import FbkError

    batch = gpu_ctx.upload([{0: D.to_fbk(oracle.OContainer.array([1]))}])
    with pytest.raises(FbkError):
        gpu_ctx.rows(batch, [0], column=1 << 20)
    with pytest.raises(FbkError):
        gpu_ctx.rows(batch, [3])
    assert gpu_ctx.rows(batch, []).size == 0
    batch.free()
