"""Pin the BSI / TopK / GroupBy part of the oracle (oracle/bsi_oracle.c) to the reference's
known answers: fragment_internal_test.go TestFragment_Sum, TestFragment_Range,
TestIntLTRegression (literals extracted into tests/golden/fragment_bsi_cases.json by
tests/golden/extract_fragment_bsi.py), the diagonal sweeps of TestFragmentBSIUnsigned /
Signed (:3768-4275, whose expectation is the true predicate over the loaded values), and
the executor-level vectors of executor_test.go (TopK :1758, Sum :2813, GroupBy :6041).
CPU only."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "fragment_bsi_cases.json")))
SW = 1 << 20


@pytest.fixture(scope="module")
def B(oracle):
    from oracle import pybsi

    pybsi._lib()
    return pybsi


@pytest.mark.parametrize("case", CASES["range_cases"], ids=lambda c: c["test"])
def test_fragment_range_golden(B, case):
    depth = max(v[1] for v in case["values"])
    frag = B.bsi_fragment_from_values({v[0]: v[2] for v in case["values"]}, depth)
    assert case["queries"], case["test"]
    for q in case["queries"]:
        if q["kind"] == "rangeOp":
            got = B.columns(B.bsi_range(frag, B.OPS[q["op"]], q["depth"], q["pred"]))
        elif q["kind"] == "rangeBetween":
            got = B.columns(B.bsi_range_between(frag, q["depth"], q["lo"], q["hi"]))
        elif q["kind"] == "rangeBetweenUnsigned":
            got = B.columns(B.bsi_range_between_unsigned(frag, B.row_from_columns(q["filter"]), q["depth"], q["lo"], q["hi"]))
        elif q["kind"] == "rangeLTUnsigned":
            got = B.columns(B.bsi_range_lt_unsigned(frag, B.row_from_columns(q["filter"]), q["depth"], q["pred"], q["allow_eq"]))
        else:
            got = B.columns(B.bsi_range_gt_unsigned(frag, B.row_from_columns(q["filter"]), q["depth"], q["pred"], q["allow_eq"]))
        assert got == q["exp"], (case["test"], q)


def test_fragment_sum_golden(B):
    sc = CASES["sum_case"]
    depth = sc["values"][0][1]
    frag = B.bsi_fragment_from_values({v[0]: v[2] for v in sc["values"]}, depth)
    for s in sc["sums"]:
        flt = B.row_from_columns(s["filter"]) if s["filter"] is not None else None
        assert B.bsi_sum(frag, flt, flt is not None) == (s["sum"], s["count"]), s
    # a filter with no contents for this shard -> (0, 0) (fragment.go:736-738)
    assert B.bsi_sum(frag, None, True) == (0, 0)


def _truth(vals, op, p):
    f = {"LT": lambda v: v < p, "LTE": lambda v: v <= p, "GT": lambda v: v > p, "GTE": lambda v: v >= p, "EQ": lambda v: v == p, "NEQ": lambda v: v != p}[op]
    return sorted(c for c, v in vals.items() if f(v))


def test_bsi_unsigned_diagonal(B):
    """TestFragmentBSIUnsigned (fragment_internal_test.go:3768-3948): k = 6, column i holds
    value i for i in [0, 64); every predicate in [-3, 2^(k+1)) for <, <=, >, >=, ==, != and
    BETWEEN."""
    k = 6
    vals = {i: i for i in range(1 << k)}
    frag = B.bsi_fragment_from_values(vals, k)
    lo_c, hi_c = -3, 1 << (k + 1)
    for p in range(lo_c, hi_c):
        for op in ("LT", "LTE", "GT", "GTE", "EQ", "NEQ"):
            assert B.columns(B.bsi_range(frag, B.OPS[op], k, p)) == _truth(vals, op, p), (op, p)
    for a in range(lo_c, hi_c, 3):
        for b in range(a, hi_c, 5):
            exp = sorted(c for c, v in vals.items() if a <= v <= b)
            assert B.columns(B.bsi_range_between(frag, k, a, b)) == exp, (a, b)


def test_bsi_signed_diagonal(B):
    """TestFragmentBSISigned (:4113-4275): k = 6, column (v + 63) holds v for v in [-63, 63];
    predicates in [-126, 126)."""
    k = 6
    mn, mx = 1 - (1 << k), (1 << k) - 1
    vals = {v - mn: v for v in range(mn, mx + 1)}
    frag = B.bsi_fragment_from_values(vals, k)
    for p in range(2 * mn, 2 * mx):
        for op in ("LT", "LTE", "GT", "GTE", "EQ", "NEQ"):
            assert B.columns(B.bsi_range(frag, B.OPS[op], k, p)) == _truth(vals, op, p), (op, p)
    for a in range(2 * mn, 2 * mx, 7):
        for b in range(a, 2 * mx, 11):
            exp = sorted(c for c, v in vals.items() if a <= v <= b)
            assert B.columns(B.bsi_range_between(frag, k, a, b)) == exp, (a, b)
    s, c = B.bsi_sum(frag, None, False)
    assert (s, c) == (sum(vals.values()), len(vals))


def test_bsi_depth64_and_wraparound(B):
    """bitDepth 64 (fragment_internal_test.go:899-915) and Go shift semantics:
    (1<<64)-1, (^0)<<64, absInt64(MinInt64)."""
    vals = {1: 0xF0, 2: 0xF1, 3: -5, 4: (1 << 62) + 12345, 5: -(1 << 62), 6: 0}
    frag = B.bsi_fragment_from_values(vals, 64)
    for p in (0, 1, -1, 0xF0, 0xF1, -5, 1 << 62, -(1 << 62), (1 << 63) - 1, -(1 << 63)):
        for op in ("LT", "LTE", "GT", "GTE", "EQ", "NEQ"):
            assert B.columns(B.bsi_range(frag, B.OPS[op], 64, p)) == _truth(vals, op, p), (op, p)
    assert B.columns(B.bsi_range_between(frag, 64, 0xF0, 0xF1)) == [1, 2]
    s, c = B.bsi_sum(frag, None, False)
    assert c == 6 and s == sum(vals.values())


def test_bsi_random_vs_truth(B):
    rng = np.random.default_rng(5)
    depth = 20
    cols = rng.choice(SW, size=3000, replace=False)
    v = rng.integers(-(1 << depth) + 1, 1 << depth, size=cols.size)
    vals = {int(c): int(x) for c, x in zip(cols, v)}
    frag = B.bsi_fragment_from_values(vals, depth)
    for p in [int(x) for x in rng.integers(-(1 << depth), 1 << depth, size=12)] + [0, 1, -1]:
        for op in ("LT", "LTE", "GT", "GTE", "EQ", "NEQ"):
            assert B.columns(B.bsi_range(frag, B.OPS[op], depth, p)) == _truth(vals, op, p), (op, p)
    fcols = [int(c) for c in cols[::3]] + [17, 99999]
    flt = B.row_from_columns(fcols)
    fs = set(fcols)
    s, c = B.bsi_sum(frag, flt, True)
    assert c == sum(1 for k in vals if k in fs) and s == sum(x for k, x in vals.items() if k in fs)


def test_executor_level_vectors(B, oracle):
    """Cross-shard vectors of executor_test.go restated per shard as bit lists.
    TopK (:1758-1809): rows {0,10,20} -> {10:4, 0:3}; Sum (:2813-2871): foo = 20,30,40,50,60 ->
    (200, 5), with Row(x=0) = cols {0, SW+1} -> (80, 2); GroupBy (:6041-6118)."""
    O = oracle
    # Sum: values over 3 shards; the reduce is ValCount.Add (executor.go:8438)
    foo = {0: 20, SW: 30, SW + 2: 40, 5 * SW + 100: 50, SW + 1: 60}
    by_shard = {}
    for col, val in foo.items():
        by_shard.setdefault(col // SW, {})[col % SW] = val
    tot = [0, 0]
    for sh, vals in by_shard.items():
        s, c = B.bsi_sum(B.bsi_fragment_from_values(vals, 8), None, False)
        tot[0] += s
        tot[1] += c
    assert tuple(tot) == (200, 5)
    filt_cols = [0, SW + 1]
    tot = [0, 0]
    for sh, vals in by_shard.items():
        fc = [c % SW for c in filt_cols if c // SW == sh]
        flt = B.row_from_columns(fc) if fc else None
        s, c = B.bsi_sum(B.bsi_fragment_from_values(vals, 8), flt, True)
        tot[0] += s
        tot[1] += c
    assert tuple(tot) == (80, 2)
    # TopK-style per-row counts with and without a filter
    rows = {0: [0, 1, 2], 10: [2, 3, 4, 5], 20: [7]}
    frag = B.Fragment([O.bitmap_from_values(rows[r]) if r in rows else None for r in range(21)])
    cnt = B.topk_row_counts(frag, None)
    assert {r: int(cnt[r]) for r in rows} == {0: 3, 10: 4, 20: 1}
    cnt = B.topk_row_counts(frag, B.row_from_columns([2, 3, 7]))
    assert {r: int(cnt[r]) for r in rows} == {0: 1, 10: 2, 20: 1}
    # GroupBy count matrix
    a = B.Fragment([O.bitmap_from_values([0, 1, 2, 70000]), O.bitmap_from_values([2, 3])])
    b = B.Fragment([O.bitmap_from_values([1, 2, 3]), O.bitmap_from_values([70000, 5]), None])
    assert B.groupby_counts(a, b, None).tolist() == [[2, 1, 0], [2, 0, 0]]
    assert B.groupby_counts(a, b, B.row_from_columns([2, 70000])).tolist() == [[1, 1, 0], [1, 0, 0]]
    # UnionRows
    assert B.columns(B.union_rows(a)) == [0, 1, 2, 3, 70000]


def val_count_smaller(a, b):
    """ValCount.Smaller for integer fields, executor.go:8446-8468; a, b = (val, count)."""
    if a[1] == 0 or (b[0] < a[0] and b[1] > 0):
        return b
    return (a[0], a[1] + (b[1] if a[0] == b[0] else 0))


def val_count_larger(a, b):
    """ValCount.Larger for integer fields, executor.go:8526-8548."""
    if a[1] == 0 or (b[0] > a[0] and b[1] > 0):
        return b
    return (a[0], a[1] + (b[1] if a[0] == b[0] else 0))


def test_fragment_minmax_golden(B):
    """TestFragment_MinMax (fragment_internal_test.go:524-604)."""
    mc = CASES["minmax_case"]
    frag = B.bsi_fragment_from_values({v[0]: v[2] for v in mc["values"]}, mc["depth"])
    for kind, fn in (("min", B.bsi_min), ("max", B.bsi_max)):
        for t in mc[kind]:
            flt = B.row_from_columns(t["filter"]) if t["filter"] is not None else None
            assert fn(frag, flt, mc["depth"]) == (t["exp"], t["cnt"]), (kind, t)


def test_minmax_executor_vectors_and_truth(B):
    """TestExecutor_Execute_MinMax (executor_test.go:2530-2660): f = {0:20, 1:-5, 2:-5, 3:10,
    SW:30, SW+2:40, 5SW+100:50, SW+1:60}; x=0 -> {0, 3, SW+1}, x=1 -> {1}, x=2 -> {SW+2};
    Min: (-5,2), (10,1), (-5,1), (40,1); Max: (60,1), (60,1), (-5,1), (40,1).  The per-shard
    results fold with ValCount.Smaller / Larger; shards without the field contribute (0, 0)."""
    f = {0: 20, 1: -5, 2: -5, 3: 10, SW: 30, SW + 2: 40, 5 * SW + 100: 50, SW + 1: 60}
    x = {0: [0, 3, SW + 1], 1: [1], 2: [SW + 2]}
    exp_min = {None: (-5, 2), 0: (10, 1), 1: (-5, 1), 2: (40, 1)}
    exp_max = {None: (60, 1), 0: (60, 1), 1: (-5, 1), 2: (40, 1)}
    depth = 11  # bitDepth of an int field with range (-1110, 1000); Base = 0
    by_shard = {}
    for col, val in f.items():
        by_shard.setdefault(col // SW, {})[col % SW] = val
    for row in (None, 0, 1, 2):
        mn, mx = (0, 0), (0, 0)
        for sh in sorted(by_shard):
            frag = B.bsi_fragment_from_values(by_shard[sh], depth)
            if row is None:
                flt = None
            else:
                # executeBitmapCallShard gives an empty Row for a shard without bits
                flt = B.row_from_columns([c % SW for c in x[row] if c // SW == sh])
            mn = val_count_smaller(mn, B.bsi_min(frag, flt, depth))
            mx = val_count_larger(mx, B.bsi_max(frag, flt, depth))
        assert mn == exp_min[row] and mx == exp_max[row], (row, mn, mx)
    # randomized: truth is min/max over the python dict
    rng = np.random.default_rng(11)
    for trial in range(30):
        depth = int(rng.integers(1, 64))
        n = int(rng.integers(1, 300))
        cols = rng.choice(1 << 20, size=n, replace=False)
        lim = (1 << depth) - 1
        kind = trial % 3
        vals = {}
        for c in cols:
            m = int(rng.integers(0, lim + 1)) if depth < 63 else int(rng.integers(0, 1 << 62))
            if kind == 0:
                v = m
            elif kind == 1:
                v = -m
            else:
                v = m if rng.random() < 0.5 else -m
            vals[int(c)] = v
        frag = B.bsi_fragment_from_values(vals, depth)
        fcols = [int(c) for c in cols[::2]] + [12345]
        for flt_cols in (None, fcols):
            sel = vals if flt_cols is None else {c: v for c, v in vals.items() if c in set(flt_cols)}
            flt = None if flt_cols is None else B.row_from_columns(flt_cols)
            tmin, tmax = min(sel.values()), max(sel.values())
            assert B.bsi_min(frag, flt, depth) == (tmin, sum(1 for v in sel.values() if v == tmin)), (trial, depth)
            assert B.bsi_max(frag, flt, depth) == (tmax, sum(1 for v in sel.values() if v == tmax)), (trial, depth)
    # depth 64 wrap: 1 << 63 as int64 is MinInt64 (Go wraps, fragment.go:795)
    frag = B.bsi_fragment_from_values({5: (1 << 63) + 3}, 64)
    assert B.bsi_max(frag, None, 64) == (-(1 << 63) + 3, 1)


def topn_vectors():
    """TestFragment_TopN_Intersect, _Intersect_Large and _IDs (fragment_internal_test.go:1174-1273):
    (row id -> columns, src columns or None, N (0 = all), expected [(id, count)], RowIDs or None)."""
    small = ({100: [1, 10, 11, 12], 101: [1, 2, 3, 4], 102: [1, 2, 4, 5, 6], 103: [1000, 1001, 1002]}, [1, 2, 3], 3, [(101, 3), (102, 2), (100, 1)], None)
    large = ({i: list(range(i)) for i in range(1000)}, list(range(980, 1000)), 10, [(999 - d, 19 - d) for d in range(10)], None)
    ids = ({100: [1, 2, 3], 101: [4, 5, 6, 7], 102: [8, 9, 10, 11, 12]}, None, 0, [(101, 4), (100, 3)], [100, 101, 200])  # row 200 does not exist
    return [small, large, ids]


def test_fragment_topn_intersect_vectors(oracle):
    from oracle import pybsi as B

    for rows, src, n, want, row_ids in topn_vectors():
        ids = sorted(rows) if row_ids is None else row_ids
        frag = B.Fragment([B.row_from_columns(rows[i]) if rows.get(i) else None for i in ids])
        cnt = B.topk_row_counts(frag, B.row_from_columns(src) if src is not None else None)
        pairs = sorted([(ids[k], int(c)) for k, c in enumerate(cnt) if c], key=lambda p: (-p[1], p[0]))
        assert (pairs[:n] if n else pairs) == want
