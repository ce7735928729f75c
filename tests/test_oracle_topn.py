"""The oracle's restatement of fragment.top against the reference's own known answers
(tests/golden/topn_vectors.json, extracted by tests/golden/extract_topn_vectors.py), and the
specification the GPU implements (top_exact) against that restatement."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
VEC = json.load(open(os.path.join(HERE, "golden", "topn_vectors.json")))["cases"]


def case_rows(c):
    if "generator" in c:
        assert c["generator"]["kind"] == "row_i_has_columns_below_i"
        return {i: list(range(i)) for i in range(c["generator"]["n"])}
    return {int(k): v for k, v in c["rows"].items()}


def test_fragment_top_reproduces_the_reference_vectors():
    from oracle import pytopn as T

    for c in VEC:
        o = c["options"]
        got = T.fragment_top(case_rows(c), o["N"], c["src"] if o["Src"] else None, o["RowIDs"], o["MinThreshold"], o["TanimotoThreshold"])
        assert [list(p) for p in got] == c["expected"], c["test"]


def test_exact_specification_agrees_with_the_restatement():
    """Without ties at the cut the rank-cache walk of fragment.top and the exhaustive rule give the
    same pairs; with N = 0 they always give the same set."""
    from oracle import pytopn as T

    for c in VEC:
        o = c["options"]
        rows = case_rows(c)
        ids = sorted(rows) if not o["RowIDs"] else o["RowIDs"]
        n = 0 if o["RowIDs"] else o["N"]
        got = T.top_exact([rows], ids, n, [c["src"]] if o["Src"] else None, o["MinThreshold"], o["TanimotoThreshold"])
        assert [list(p) for p in got] == c["expected"], c["test"]
    rng = np.random.default_rng(5)
    for trial in range(200):
        rows = {int(r): sorted(set(rng.integers(0, 40, int(rng.integers(0, 25))).tolist())) for r in range(12)}
        src = sorted(set(rng.integers(0, 40, int(rng.integers(1, 20))).tolist())) if trial % 3 else None
        mt = int(rng.integers(0, 6)) if trial % 2 else 0
        tt = int(rng.choice([0, 10, 30, 50, 80])) if src is not None and trial % 5 == 0 else 0
        a = T.fragment_top(rows, 0, src, None, mt, tt)
        b = T.top_exact([rows], sorted(rows), 0, [src] if src is not None else None, mt, tt)
        assert sorted(a) == sorted(b), (trial, rows, src, mt, tt)
        assert [p[1] for p in a] == [p[1] for p in b]  # both are in descending count order


def _shards_of(bits, width):
    """[(row, column)] -> one {row: [columns relative to the shard]} per shard, from shard 0 to the last one that has a bit"""
    n = max(c // width for _, c in bits) + 1
    out = [dict() for _ in range(n)]
    for r, c in bits:
        out[c // width].setdefault(r, []).append(c % width)
    return out


def test_execute_topn_against_the_reference_executor_vectors():
    """oracle/pytopn.execute_topn (executeTopN, executor.go:2779-2864: per-SHARD fragment.top candidates, merged untrimmed,
    re-counted, trimmed) reproduces TestExecutor_Execute_TopN / _fill / _fill_small / _Src (executor_test.go:1846-2200),
    extracted by tests/golden/extract_executor_topn.py.  _fill_small is the cross-shard case: five shards whose own top-1
    rows are 1, 2, 3, 4 and 0 -> candidates {0..4} -> {0: 5}."""
    from oracle import pytopn as T

    g = json.load(open(os.path.join(GOLD, "executor_topn_vectors.json")))
    w = g["shard_width"]
    assert [c["test"] for c in g["cases"]] == ["TestExecutor_Execute_TopN/RowIDColumnID", "TestExecutor_Execute_TopN_fill", "TestExecutor_Execute_TopN_fill_small",
                                                "TestExecutor_Execute_TopN_Src"]
    for c in g["cases"]:
        shards = _shards_of([tuple(b) for b in c["bits"]], w)
        srcs = None
        if c["src_bits"] is not None:
            srcs = [[x % w for x in c["src_bits"] if x // w == s] for s in range(len(shards))]
        got = T.execute_topn(shards, c["n"], srcs)
        assert [list(p) for p in got] == c["expected"], (c["test"], got)
    fs = [c for c in g["cases"] if c["test"].endswith("fill_small")][0]
    assert T.topn_candidates(_shards_of([tuple(b) for b in fs["bits"]], w), 1) == [0, 1, 2, 3, 4]


def test_execute_topn_is_neither_exact_nor_per_node():
    """A constructed input on which the three candidate rules differ: the reference's (per shard), the exact top n, and
    round 4's mistaken per-node rule (the first n of a node's MERGED totals)."""
    from oracle import pytopn as T

    # shards 0 and 1 live on one node, shard 2 on another.  Row 2 is second in every shard and has the largest total.
    s0 = {0: range(10), 1: range(1), 2: range(9)}
    s1 = {0: range(1), 1: range(10), 2: range(9)}
    s2 = {3: range(5), 2: range(4)}
    ids = [0, 1, 2, 3]
    assert T.top_exact([s0, s1, s2], ids, 1) == [(2, 22)]
    # per shard: top-1 rows are 0, 1, 3 -> candidates {0, 1, 3}; totals 11, 11, 5 -> (0, 11)
    assert T.topn_candidates([s0, s1, s2], 1) == [0, 1, 3]
    assert T.execute_topn([s0, s1, s2], 1) == [(0, 11)]
    # (the per-node rule would have merged s0 + s1 first: totals 0: 11, 1: 11, 2: 18 -> candidate 2 -> (2, 22): not the reference's answer)
    assert T.execute_topn([s0, s1, s2], 2) == [(2, 22), (0, 11)]  # n = 2: every shard's first two rows -> all four rows
    assert T.execute_topn([s0, s1, s2], 0) == T.top_exact([s0, s1, s2], ids, 0)
    # ids given: pass 1 is the answer, untrimmed (executor.go:2800-2806)
    assert T.execute_topn([s0, s1, s2], 1, ids_arg=[1, 2]) == [(2, 22), (1, 11)]
    # with a source row a shard can return MORE than n pairs: rows after the n-th whose count reaches the heap's minimum
    # are pushed without evicting (fragment.go:1404-1425)
    rows = {0: range(0, 10), 1: range(5, 14), 2: range(6, 14), 3: range(100, 101)}
    src = range(5, 20)
    assert T.fragment_top(rows, 1, src, None, 1, 0) == [(1, 9), (2, 8), (0, 5)]
    assert T.execute_topn([rows], 1, [src]) == [(1, 9)]


def test_execute_topn_random_against_the_candidate_definition():
    """execute_topn == (candidates = union of the shards' fragment.top(n) ids) then top_exact over the candidates, trimmed."""
    from oracle import pytopn as T

    rng = np.random.default_rng(11)
    for trial in range(150):
        ns = int(rng.integers(1, 5))
        shards = [{int(r): sorted(set(rng.integers(0, 50, int(rng.integers(0, 30))).tolist())) for r in range(10)} for _ in range(ns)]
        srcs = [sorted(set(rng.integers(0, 50, int(rng.integers(1, 25))).tolist())) for _ in range(ns)] if trial % 3 else None
        mt = int(rng.integers(0, 5)) if trial % 2 else 0
        tt = int(rng.choice([0, 10, 30, 50])) if srcs is not None and trial % 4 == 0 else 0
        n = int(rng.integers(1, 6))
        cand = T.topn_candidates(shards, n, srcs, mt, tt)
        exp = T.top_exact(shards, cand, n, srcs, max(mt, 1), tt)
        assert T.execute_topn(shards, n, srcs, None, mt, tt) == exp, trial


def _literal_float64_rule(cnt, count, src_count, has_src, min_threshold, tanimoto_threshold):
    """fragment.go:1334-1385 as written: float64 comparisons and math.Ceil (numpy float64 = Go float64, IEEE 754 binary64)."""
    if cnt == 0 or count == 0:
        return False
    if tanimoto_threshold > 0 and has_src:
        min_t = np.float64(src_count * tanimoto_threshold) / np.float64(100)
        max_t = np.float64(src_count * 100) / np.float64(tanimoto_threshold)
        if np.float64(cnt) <= min_t or np.float64(cnt) >= max_t:
            return False
        t = np.ceil(np.float64(count * 100) / np.float64(cnt + src_count - count))
        return not (t <= np.float64(tanimoto_threshold))
    return not (cnt < min_threshold) and not (count < min_threshold)


def test_integer_rule_equals_the_float64_rule_at_shard_scale_counts():
    """oracle/pytopn.row_passes (and the device's TopnRule, fbk_query_kernels.hip.h) replace fragment.top's float64 comparisons
    (fragment.go:1334-1385) by integer ones.  They are the same predicate for every count a shard can hold: 10^5 random
    (cnt, count, src_count, threshold) with counts up to 2^20, and every neighbour of the boundaries cnt * 100 = src * t,
    cnt * t = src * 100 and count * 100 = t * (cnt + src - count) that an integer triple can reach."""
    from oracle import pytopn as T

    rng = np.random.default_rng(0x70b9)
    n = 100_000
    W = 1 << 20
    cnts = rng.integers(1, W + 1, n)
    srcs = rng.integers(1, W + 1, n)
    counts = np.minimum(np.minimum(cnts, srcs), rng.integers(0, W + 1, n))
    counts = np.maximum(counts, cnts + srcs - W).clip(0)  # |row ∪ src| <= 2^20
    tts = rng.choice([1, 3, 10, 25, 30, 33, 50, 66, 70, 90, 99, 100], n)
    mts = rng.integers(0, W, n)
    for i in range(n):
        args = (int(cnts[i]), int(counts[i]), int(srcs[i]), True, 0, int(tts[i]))
        assert T.row_passes(*args) == _literal_float64_rule(*args), args
        args = (int(cnts[i]), int(counts[i]), int(srcs[i]), bool(i & 1), int(mts[i]) if i % 3 else int(cnts[i]), 0)
        assert T.row_passes(*args) == _literal_float64_rule(*args), args
    # the boundaries themselves and their neighbours
    checked = 0
    for t in (1, 7, 10, 30, 50, 75, 99, 100):
        for src in (1, 3, 100, 1000, 4097, 65536, 333_333, 999_999, W):
            for cnt0 in {src * t // 100, -(-src * t // 100), src * 100 // t, -(-src * 100 // t)}:
                for cnt in (cnt0 - 1, cnt0, cnt0 + 1):
                    if not 1 <= cnt <= W:
                        continue
                    lo, hi = max(0, cnt + src - W), min(cnt, src)
                    # count * 100 = t * (cnt + src - count)  <=>  count = t (cnt + src) / (100 + t)
                    c0 = t * (cnt + src) // (100 + t)
                    for count in {lo, hi, (lo + hi) // 2, c0 - 1, c0, c0 + 1, c0 + 2}:
                        if lo <= count <= hi:
                            args = (cnt, count, src, True, 0, t)
                            assert T.row_passes(*args) == _literal_float64_rule(*args), args
                            checked += 1
    assert checked > 1500
