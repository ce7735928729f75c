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
