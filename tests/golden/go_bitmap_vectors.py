"""The reference's bitmap-level known answers for IntersectionCount, restated as data:
TestBitmap_IntersectionCount_{ArrayArray, ArrayRun, RunRun, BitmapRun, ArrayBitmap, BitmapBitmap,
Mixed} (roaring/roaring_test.go:1283-1387) and testBM() (:1661-1684, "count 75007").

A bitmap is (values, optimized): roaring.NewFileBitmap(values...) grows containers by Add — an
array until it holds 4096 values, a bitmap from then on (arrayAdd, roaring.go:3248-3277) — and
`Optimize()` re-encodes every container by Container.optimize() (roaring.go:3412-3461)."""


def test_bm_values():
    v = [(1 << 16) + i for i in range(0, 1024, 4)]  # the array
    v += [(2 << 16) + i for i in range(0, 16384, 2)]  # the bitmap
    v += [(3 << 16) + i for i in range(1024)]  # small run
    v += [(4 << 16) + i for i in range(65535)]  # large run
    return v


TEST_BM = (test_bm_values(), True)
TEST_BM_COUNT = 75007

# (name, bm0, bm1, expected |bm0 ∩ bm1|); every case is also checked in reverse
CASES = [
    ("ArrayArray", ([0, 1000001, 1000002, 1000003], False), ([0, 50000, 999998, 999999, 1000000, 1000001, 1000002], False), 3),
    ("ArrayRun", ([0, 1000001, 1000002, 1000003], False), ([0, 1, 2, 3, 4, 5, 1000000, 1000002, 1000003, 1000004, 1000005, 1000006], True), 3),
    ("RunRun", ([3, 4, 5, 6, 7, 8, 1000001, 1000002, 1000003, 1000004], True),
     ([0, 1, 2, 3, 4, 5, 1000000, 1000002, 1000003, 1000004, 1000005, 1000006], True), 6),
    ("BitmapRun", (list(range(3, 1000007, 2)), False), ([0, 1, 2, 3, 4, 5, 1000000, 1000002, 1000003, 1000004, 1000005, 1000006], True), 4),
    ("ArrayBitmap", ([1, 70, 200, 4097, 4098], False), (list(range(0, 10001, 2)), False), 3),
    ("BitmapBitmap", (sorted(set(list(range(0, 10001, 2)) + [1000, 2000])), False), (sorted(set(list(range(1, 10002, 2)) + [1000, 2000])), False), 2),
    ("Mixed/self", TEST_BM, TEST_BM, TEST_BM_COUNT),
    ("Mixed/1", TEST_BM, ([0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 65536], False), 1),
    ("Mixed/3", TEST_BM, ([131072], False), 1),
]


def _run_bitmap(run_len, space_len, offset, limit_sub):
    v = []
    i = 0
    while i < 65536 - run_len - limit_sub:
        v.extend(offset + i + j for j in range(run_len))
        i += run_len + space_len
    return v


_BIG = list(range(628, 2683301))
_EVEN10K = list(range(0, 10000, 2))
SW = 1 << 20

# (name, op, bm0, bm1, expected count, expected Slice() or None): TestBitmap_Intersection (:483),
# _Union1 (:781), _Intersection_Empty (:942), _IntersectArrayArray (:953), _IntersectBitmapBitmap
# (:979), _IntersectRunRun (:1000), _Difference* (:1041-1089), _Union (:1091), _Xor* (:1124-1216)
SETOP_CASES = [
    ("Intersection", "and", ([0, 2683177], False), (_BIG, False), 1, None),
    ("Union1/a", "or", ([0, 2683177], False), (_BIG + [4000000], False), 2682675, None),
    ("Union1/testBM", "or", TEST_BM, ([0, 2683177], False), 75009, None),
    ("Union1/self", "or", TEST_BM, TEST_BM, 75007, None),
    ("Intersection_Empty", "and", ([0, 2683177], False), ([], False), 0, []),
    ("IntersectArrayArray", "and", ([0, 1, 7, 9, 11, 2683, 5005], False), ([0, 2683, 2684, 5000], False), 2, [0, 2683]),
    ("IntersectArrayArray/rev", "and", ([0, 2683, 2684, 5000], False), ([0, 1, 7, 9, 11, 2683, 5005], False), 2, [0, 2683]),
    ("IntersectBitmapBitmap", "and", (list(range(0, 65536, 2)), False), (list(range(0, 65536, 3)), False), 10923, None),
    ("IntersectRunRun/array", "and", ([0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15], True), ([5, 6, 7, 8, 9, 10, 11], True), 3, [5, 10, 11]),
    ("IntersectRunRun/bitmap", "and", (_run_bitmap(25, 8, 25 // 2 + 8, 25 // 2 + 8), True), (_run_bitmap(32, 1, 0, 0), True), 47628, None),
    ("Difference", "andnot", ([0, 2683177], False), (_BIG, False), 1, [0]),
    ("Difference2", "andnot", ([0, 1, 2, 131072, 262144, SW + 5, SW + 7], False), ([2, 3, 100000, 262144, 2 * SW + 1], False), 5,
     [0, 1, 131072, SW + 5, SW + 7]),
    ("Difference_Empty", "andnot", ([0, 2683177], False), ([], False), 2, [0, 2683177]),
    ("DifferenceArrayArray", "andnot", ([0, 4, 8, 12, 16, 20], False), ([1, 3, 6, 9, 12, 15, 18], False), 5, None),
    ("DifferenceArrayRun", "andnot", ([0, 4, 8, 12, 16, 20, 36, 40, 44], False), ([1, 2, 3, 4, 5, 6, 7, 8, 9, 30, 31, 32, 33, 34, 35, 36], True), 6, None),
    ("Union", "or", ([0, 1000001, 1000002, 1000003], False), ([0, 50000, 1000001, 1000002], False), 5, None),
    ("Xor/a", "xor", ([0, 1, 2, 3], False), TEST_BM, 75011, None),
    ("Xor/b", "xor", TEST_BM, ([0, 1, 2, 3], False), 75011, None),
    ("Xor/self", "xor", TEST_BM, TEST_BM, 0, []),
    ("Xor_ArrayArray", "xor", ([0, 1000001, 1000002, 1000003], False), ([0, 50000, 1000001, 1000002], False), 2, [50000, 1000003]),
    ("Xor_Empty", "xor", ([0, 50000, 1000001, 1000002], False), ([], False), 4, None),
    ("Xor_ArrayBitmap", "xor", ([1, 70, 200, 4097, 4098], False), (_EVEN10K, False), 4999, None),
    ("Xor_ArrayBitmap/rev", "xor", (_EVEN10K, False), ([1, 70, 200, 4097, 4098], False), 4999, None),
    ("Xor_ArrayBitmap/empty", "xor", (_EVEN10K, False), ([], False), 5000, None),
    ("Xor_BitmapBitmap", "xor", (list(range(1, 10000, 2)), False), (_EVEN10K, False), 10000, None),
]


def _edge_case_values():
    # TestBitmap_BitmapCountRangeEdgeCase (roaring_test.go:368-390)
    start = 2009 * SW + (39314024 % SW)
    v = []
    for i in range(65536):
        start += 16384 if (i + 1) % 4096 == 0 else 2
        v.append(start)
    return v


_RUNS23 = [0, 1, 2, 3, 4, 5, 12, 13, 14, 15, 16, 17, 1000000, 1000002, 1000003, 1000004, 1000005, 1000006, 1000010, 1000011, 1000012, 1000013, 1000014]

# (name, bitmap, [(start, end, expected CountRange)]): TestBitmap_BitmapCountRangeEdgeCase (:368),
# _BitmapCountRange (:392), _ArrayCountRange (:429), _RunCountRange (:457).  The reference's
# start > end cases return 0 (and panic under roaringSentinel).
COUNT_RANGE_CASES = [
    ("EdgeCase", (_edge_case_values(), False), [(2009 * SW, 2010 * SW, 65536)]),
    ("BitmapCountRange", ([0, 2683177] + _BIG + [2683307], False),
     [(1, 2683311, 2682674), (2683177, 2683310, 125), (2683301, 3000000, 1), (0, 1, 1), (10000000, 10000001, 0), (65536, 2, 0)]),
    ("ArrayCountRange", ([0, 2683177, 2683313], False), [(1, 2683313, 1), (2621440, 2, 0)]),
    ("RunCountRange/0", (_RUNS23, True), [(15, 1000003, 5)]),
    ("RunCountRange/1", (list(range(18)), True), [(5, 12, 7)]),
    ("RunCountRange/2", (list(range(65536, 65554)), True), [(3, 2, 0)]),
    ("RunCountRange/3", ([1, 2, 3, 4], True), [(1, 3, 2)]),
]


# TestBitmap_Intersect*InPlace (roaring_test.go:499-780): the in-place forms (intersectInPlace,
# roaring.go:855-1269) must leave the same contents as the allocating ones — same tuple layout as
# SETOP_CASES, every one an intersection.
_A7 = [0, 1, 7, 9, 11, 2683, 5005]
_THIRDS1 = list(range(1, 65536, 3))
_R25 = _run_bitmap(25, 8, 25 // 2 + 8, 25 // 2 + 8)
_R32 = _run_bitmap(32, 1, 0, 0)
INPLACE_CASES = [
    ("IntersectionInPlace", "and", ([0, 2683177], False), (_BIG, False), 1, None),
    ("IntersectionInPlace_Empty/a", "and", ([0, 2683177], False), ([], False), 0, []),
    ("IntersectionInPlace_Empty/b", "and", ([], False), ([0, 2683177], False), 0, []),
    ("IntersectArrayBitmapInPlace", "and", (_A7, False), (_THIRDS1, False), 4, [1, 7, 2683, 5005]),
    ("IntersectArrayRunInPlace", "and", (_A7, False), ([5, 6, 7, 8, 9, 10, 11, 13], True), 3, [7, 9, 11]),
    ("IntersectBitmapArrayInPlace", "and", (_THIRDS1, False), (_A7, False), 4, [1, 7, 2683, 5005]),
    ("IntersectBitmapRunInPlace", "and", (_R32, False), (_R25, True), 47628, None),
    ("IntersectRunRunInPlace/array", "and", ([0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15], True), ([5, 6, 7, 8, 9, 10, 11, 13], True), 4, [5, 10, 11, 13]),
    ("IntersectRunRunInPlace/bitmap", "and", (_R25, True), (_R32, True), 47628, None),
    ("IntersectRunArrayInPlace", "and", ([0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15], True), ([5, 6, 7, 8, 9, 10, 11, 13], False), 4, [5, 10, 11, 13]),
    ("IntersectRunBitmapInPlace", "and", (_R25, True), (_R32, False), 47628, None),
]
SETOP_CASES += INPLACE_CASES

# bm0.IntersectInPlace(bm11, bm12) (roaring_test.go:640-654): a three-way fold
_TB_ARRAY = ([(1 << 16) + i for i in range(0, 1024, 4)], True)
_TB_BITMAP = ([(2 << 16) + i for i in range(0, 16384, 2)], True)
_TB_SMALLRUN = ([(3 << 16) + i for i in range(1024)], True)
_TB_LARGERUN = ([(4 << 16) + i for i in range(65535)], True)
# (name, op, [bm0, others...], expected count, expected Slice() or None): the first bitmap folded
# with the others left to right — IntersectInPlace(bm11, bm12) (roaring_test.go:640-654),
# UnionInPlace1 (:808-846: result starts empty), DifferenceInPlace (:2051-2091)
FOLD_CASES = [
    ("IntersectInPlace(bm11, bm12)", "and", [(_A7, False), ([5, 6, 7, 8, 9, 10, 11, 13, 2683], False), ([6, 7, 10, 13, 2683], False)], 2, [7, 2683]),
    ("UnionInPlace1/a", "or", [([], False), ([0, 2683177], False), (_BIG + [4000000], False)], 2682675, None),
    ("UnionInPlace1/testBM", "or", [([], False), TEST_BM, ([0, 2683177], False)], 75009, None),
    ("UnionInPlace1/self", "or", [([], False), TEST_BM, TEST_BM], 75007, None),
    ("DifferenceInPlace/all", "andnot", [TEST_BM, _TB_ARRAY, _TB_BITMAP, _TB_SMALLRUN, _TB_LARGERUN], 0, []),
    ("DifferenceInPlace/but-array", "andnot", [TEST_BM, _TB_BITMAP, _TB_SMALLRUN, _TB_LARGERUN], 256, [(1 << 16) + i for i in range(0, 1024, 4)]),
]
