"""The oracle's Bitmap.IntersectionCount / Count against the reference's bitmap-level known
answers (tests/golden/go_bitmap_vectors.py: roaring_test.go:1283-1387, testBM() :1661-1684)."""
import sys, os

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import go_bitmap_vectors as V  # noqa: E402


def file_bitmap(O, values, optimized):
    """NewFileBitmap(values...) [+ Optimize()] as (key, oracle container) pairs."""
    v = np.unique(np.asarray(values, dtype=np.uint64))
    out = []
    if v.size == 0:
        return out
    keys = v >> np.uint64(16)
    starts = np.concatenate([[0], np.nonzero(np.diff(keys))[0] + 1, [v.size]])
    for a, b in zip(starts[:-1], starts[1:]):
        lo = (v[a:b] & np.uint64(0xFFFF)).astype(np.uint16)
        if lo.size < 4096:
            c = O.OContainer.array(lo)
        else:
            w = np.zeros(65536, dtype=np.uint8)
            w[lo] = 1
            c = O.OContainer.bitmap(np.packbits(w, bitorder="little").view(np.uint64), int(lo.size))
        out.append((int(keys[a]), O.optimize(c) if optimized else c))
    return out


def test_testbm_count_and_encodings(oracle):
    O = oracle
    items = file_bitmap(O, *V.TEST_BM)
    bm = O.OBitmap.from_containers(items)
    assert bm.count() == V.TEST_BM_COUNT
    # "the array", "the bitmap", "small run", "large run" (roaring_test.go:1665-1681)
    assert [(k, c.typ) for k, c in items] == [(1, O.ARRAY), (2, O.BITMAP), (3, O.RUN), (4, O.RUN)]


@pytest.mark.parametrize("name,a,b,want", V.CASES, ids=[c[0] for c in V.CASES])
def test_bitmap_intersection_count_vectors(oracle, name, a, b, want):
    O = oracle
    A = O.OBitmap.from_containers(file_bitmap(O, *a))
    B = O.OBitmap.from_containers(file_bitmap(O, *b))
    assert A.intersection_count(B) == want
    assert B.intersection_count(A) == want  # "unexpected n (reverse)"


@pytest.mark.parametrize("name,op,a,b,want,want_slice", V.SETOP_CASES, ids=[c[0] for c in V.SETOP_CASES])
def test_bitmap_setop_vectors(oracle, name, op, a, b, want, want_slice):
    """TestBitmap_Intersection / _Union1 / _Intersect* / _Difference* / _Union / _Xor*
    (roaring_test.go:483-1216): result cardinalities and, where the reference checks them, contents."""
    O = oracle
    A = O.OBitmap.from_containers(file_bitmap(O, *a))
    B = O.OBitmap.from_containers(file_bitmap(O, *b))
    r = {"and": lambda: A.intersect(B), "or": lambda: A.union(B), "andnot": lambda: A.difference(B), "xor": lambda: A.xor(B)}[op]()
    assert r.count() == want
    if want_slice is not None:
        assert r.slice() == want_slice


@pytest.mark.parametrize("name,spec,ranges", V.COUNT_RANGE_CASES, ids=[c[0] for c in V.COUNT_RANGE_CASES])
def test_bitmap_count_range_vectors(oracle, name, spec, ranges):
    """TestBitmap_BitmapCountRangeEdgeCase / _BitmapCountRange / _ArrayCountRange / _RunCountRange
    (roaring_test.go:368-482)."""
    O = oracle
    bm = O.OBitmap.from_containers(file_bitmap(O, *spec))
    if name == "EdgeCase":
        assert bm.count() == ranges[0][2]  # "Counts != CountRange"
    for s, e, want in ranges:
        got = bm.count_range(s, e) if s <= e else 0  # start > end: the reference's loop counts nothing
        assert got == want, (s, e)


@pytest.mark.parametrize("name,op,specs,want,want_slice", V.FOLD_CASES, ids=[c[0] for c in V.FOLD_CASES])
def test_bitmap_fold_vectors(oracle, name, op, specs, want, want_slice):
    """bm0.IntersectInPlace(bm11, bm12) (roaring_test.go:640-654), UnionInPlace1 (:808-846),
    DifferenceInPlace (:2051-2091): n-way folds."""
    O = oracle
    bms = [O.OBitmap.from_containers(file_bitmap(O, *s)) for s in specs]
    if op == "or":
        r = bms[0].union(*bms[1:])
    elif op == "andnot":
        r = bms[0].difference(*bms[1:])
    else:
        r = bms[0]
        for b in bms[1:]:
            r = r.intersect(b)
    assert r.count() == want
    if want_slice is not None:
        assert r.slice() == want_slice


def test_run_count_range_overcount_is_pinned(oracle):
    """RunCountRange (roaring.go:3200-3232) counts a run twice when it starts inside the range and its
    LAST value equals `end`: the "subset of range" branch (Last <= end) adds the whole run and the
    "overlaps end" branch (Last >= end) adds end - Start on top.  The oracle restates the function as
    written, so it reproduces the over-count; fbk_count_range counts bits (include/fbk.h) — the
    deliberate divergence is pinned here instead of assumed.  Container-aligned ranges, the only
    ones the reference's callers produce (fragment.go:237,396,444), never meet the quirk."""
    import ctypes as C

    import numpy as np

    O = oracle
    runs = np.array([[10, 20], [30, 40]], dtype=np.uint16)  # bits 10..20 and 30..40
    f = O.lib().orc_run_count_range
    cnt = lambda s, e: int(f(runs.ctypes.data_as(C.c_void_p), 2, s, e))  # noqa: E731
    bits = np.zeros(65536, dtype=np.int64)
    bits[10:21] = 1
    bits[30:41] = 1
    # the quirk: [15, 40): true count = 6 + 10 = 16; the run [30, 40] has Start > start and Last == end
    assert int(bits[15:40].sum()) == 16
    assert cnt(15, 40) == 6 + 11 + 10  # subset branch adds 11 (positions 30..40), overlap branch 40 - 30 = 10 more
    # away from the quirk the function counts bits
    for s, e in [(0, 65536), (0, 10), (10, 21), (12, 35), (21, 30), (35, 36), (41, 50), (0, 41)]:
        assert cnt(s, e) == int(bits[s:e].sum()), (s, e)
    # same through Bitmap.CountRange on a one-container bitmap
    bm = O.OBitmap.from_containers([(0, O.OContainer.run([(10, 20), (30, 40)]))])
    assert bm.count_range(15, 40) == 27 and bm.count_range(12, 35) == int(bits[12:35].sum())
