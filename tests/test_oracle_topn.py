"""The oracle's restatement of fragment.top against the reference's own known answers
(tests/golden/topn_vectors.json, extracted by tests/golden/extract_topn_vectors.py), and the
specification the GPU implements (top_exact) against that restatement."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
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


def test_two_pass_topn_restatement():
    """oracle/pytopn.top_two_pass (executeTopN, executor.go:2779-2827): with one node or n = 0 it is top_exact; with several
    nodes a row that is in no node's own first n is lost — the reference's known approximation, reproduced on purpose."""
    from oracle import pytopn as T

    # node 0 ranks row 0 first, node 1 ranks row 1 first; row 2 is second on both and has the largest total
    n0 = [{0: range(10), 1: range(1), 2: range(9)}]
    n1 = [{0: range(1), 1: range(10), 2: range(9)}]
    ids = [0, 1, 2]
    assert T.top_exact(n0 + n1, ids, 1) == [(2, 18)]
    assert T.top_two_pass([n0, n1], ids, 1) == [(0, 11)]  # candidates {0, 1}: totals 11, 11 -> id ascending
    assert T.top_two_pass([n0, n1], ids, 2) == [(2, 18), (0, 11)]
    assert T.top_two_pass([n0, n1], ids, 0) == T.top_exact(n0 + n1, ids, 0)
    assert T.top_two_pass([n0 + n1], ids, 1) == T.top_exact(n0 + n1, ids, 1)
    assert T.top_two_pass([n0, []], ids, 2) == T.top_exact(n0, ids, 2)
