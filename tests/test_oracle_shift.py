"""The Shift restatement (oracle/pyshift.py) against the reference's own vectors:
TestBitmap_Shift (roaring/roaring_test.go:1389-1417) and TestExecutor_Execute_Shift
(executor_test.go:6590-6676)."""
import numpy as np
import pytest

SW = 1 << 20
MAXU64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def S(oracle):
    from oracle import pyshift

    return pyshift


def _bitmap(O, values):
    groups = {}
    for v in values:
        groups.setdefault(v >> 16, []).append(v & 0xFFFF)
    return [(k, O.optimize(O.OContainer.array(sorted(lo)))) for k, lo in sorted(groups.items())]


def _slice(items):
    return [(k << 16) + v for k, c in items for v in c.values()]


@pytest.mark.parametrize(
    "src,exp",
    [
        ([0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 65536, MAXU64], [1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 65537]),
        ([65535, 131073], [65536, 131074]),
        ([65535, 131073, 65536 * 5 - 1, 65536 * 10, 65536 * 15 - 1], [65536, 131074, 65536 * 5, 65536 * 10 + 1, 65536 * 15]),
    ],
)
def test_bitmap_shift_vectors(oracle, S, src, exp):
    assert _slice(S.bitmap_shift(_bitmap(oracle, src))) == exp


@pytest.mark.parametrize("typ", ["array", "bitmap", "run"])
def test_container_shift_every_encoding(oracle, S, typ):
    """the three per-encoding kernels agree on content and carry (values incl. 0 and 65535)"""
    O = oracle
    rng = np.random.default_rng(11)
    for top in (False, True):
        vals = set(rng.choice(65535, size=3000, replace=False).tolist()) | {0, 7, 8, 9}
        if top:
            vals |= {65534, 65535}
        words = np.zeros(1024, dtype=np.uint64)
        for v in vals:
            words[v >> 6] |= np.uint64(1) << np.uint64(v & 63)
        c = O.OContainer.from_words(words, {"array": O.ARRAY, "bitmap": O.BITMAP, "run": O.RUN}[typ])
        o, carry = S.shift_container(c)
        assert carry == top
        assert o.values() == sorted(v + 1 for v in vals if v != 65535)
        assert o.n == len(vals) - int(top)


def test_executor_shift_vectors(oracle, S):
    row = S.row_from_columns([0])  # "Shift Bit 0" executor_test.go:6592-6609
    assert S.row_columns(S.row_shift(row, 1)) == [1]
    assert S.row_columns(S.row_shift(S.row_shift(row, 1), 1)) == [2]
    row = S.row_from_columns([65535])  # "Shift container boundary" :6611-6622
    assert S.row_columns(S.row_shift(row, 1)) == [65536]
    row = S.row_from_columns([1, SW - 1, SW + 1])  # "Shift shard boundary" :6624-6654
    assert S.row_columns(S.row_shift(row, 1)) == [2, SW, SW + 2]
    assert S.row_columns(S.row_shift(row, 2)) == [3, SW + 1, SW + 3]
    row = S.row_from_columns([SW - 2, SW - 1, SW, SW + 2])  # "no create" :6656-6676
    assert S.row_columns(S.row_shift(row, 1)) == [SW - 1, SW, SW + 1, SW + 3]
    assert S.row_columns(S.row_shift(S.row_shift(row, 1), 1)) == [SW, SW + 1, SW + 2, SW + 4]
