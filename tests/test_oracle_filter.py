"""The oracle's restatement of the bitmap filter protocol (oracle/pyfilter.py <- roaring/filter.go) against the
expectations of the reference's own tests.  Those tests generate their data and expectations with loops; the
loops are restated here, each next to its citation."""
import numpy as np
import pytest

from oracle import pyfilter as PF

SAMPLE_ROWS = 100  # sampleDataSize, roaring/filter_internal_test.go:22


def sample_data():
    """requireSampleData (filter_internal_test.go:24-40): for each container offset i = 1..15, bit i of that
    container in rows 0, i, 2i, ... < 100."""
    c = {}
    for i in range(1, PF.ROW_WIDTH):
        for row in range(0, SAMPLE_ROWS, i):
            c.setdefault(row * PF.ROW_WIDTH + i, set()).add(i)
    return {k: PF.SetContainer(v) for k, v in c.items()}


def get_rows(data, *filters):  # getRows, filter_internal_test.go:45-59
    return PF.fragment_rows(data, 0, filters)


def test_base_filter():  # TestBaseFilter, filter_internal_test.go:76-85: every row (stride 1 hits them all)
    assert get_rows(sample_data()) == list(range(SAMPLE_ROWS))


@pytest.mark.parametrize("i", range(1, PF.ROW_WIDTH))
def test_column_filter(i):  # TestColumnFilter, filter_internal_test.go:87-100
    base = (i << 16) + i
    assert get_rows(sample_data(), PF.ColumnFilter(base)) == list(range(0, SAMPLE_ROWS, i))


def test_rows_filter():  # TestRowsFilter, filter_internal_test.go:102-115
    row_set = [0, 1, 2, 3]
    expected = [0, 2]
    assert get_rows(sample_data(), PF.RowsFilter(row_set)) == row_set
    assert get_rows(sample_data(), PF.RowsFilter(row_set), PF.ColumnFilter((2 << 16) + 2)) == expected
    assert get_rows(sample_data(), PF.ColumnFilter((2 << 16) + 2), PF.RowsFilter(row_set)) == expected
    assert get_rows(sample_data(), PF.ColumnFilter((2 << 16) + 2), PF.RowsFilter(row_set), PF.RowLimitFilter(1)) == expected[:1]


def test_fragment_rows_iteration_first_container():  # TestFragment_RowsIteration/firstContainer, fragment_internal_test.go:3018-3046
    data = {i * 16: PF.SetContainer([i % 2]) for i in range(100, 200)}
    assert PF.fragment_rows(data) == list(range(100, 200))
    assert PF.fragment_rows(data, 0, [PF.ColumnFilter(1)]) == [i for i in range(100, 200) if i % 2 == 1]


def test_fragment_rows_iteration_second_row():  # .../secondRow, fragment_internal_test.go:3048-3080
    data = {1 * 16 + 1: PF.SetContainer([66000 & 0xFFFF]), 2 * 16 + 1: PF.SetContainer([66000 & 0xFFFF]), 2 * 16 + 2: PF.SetContainer([166000 & 0xFFFF])}
    assert PF.fragment_rows(data) == [1, 2]
    assert PF.fragment_rows(data, 0, [PF.ColumnFilter(66000)]) == [1, 2]


def test_fragment_rows_iteration_combinations():  # .../combinations, fragment_internal_test.go:3082-3110 (every 8th step checked)
    data, expected = {}, []
    step = 0
    for r in range(1, 10000, 250):
        expected.append(r)
        for c in range(1, (1 << 20) - 1, (1 << 20) >> 5):
            data.setdefault(r * 16 + (c >> 16), PF.SetContainer([])).v.add(c & 0xFFFF)
            step += 1
            if step % 8 == 0 or c == 1:
                assert PF.fragment_rows(data) == expected
                assert PF.fragment_rows(data, 0, [PF.ColumnFilter(c)]) == expected


def test_skip_ahead_is_used():
    """The point of the protocol (filter.go:181-192): a column filter looks at ONE container per row; everything
    else is skipped by key alone."""
    rng = np.random.default_rng(7)
    data = {r * 16 + s: PF.SetContainer(rng.integers(0, 65536, 5)) for r in range(300) for s in range(16)}
    col = (9 << 16) + 77
    for r in (5, 17, 123):
        data[r * 16 + 9].v.add(77)
    stats = {}
    assert PF.fragment_rows(data, 0, [PF.ColumnFilter(col)], stats) == sorted(r for r in range(300) if data[r * 16 + 9].contains(77))
    assert stats["consider_data"] == 300  # one container per row had to be opened
    assert stats["consider_key"] <= 2 * 300 + 16 and stats["skipped"] >= 300 * 13


def test_start_row_and_limit():
    data = {r * 16 + (r % 16): PF.SetContainer([3]) for r in range(0, 400, 3)}
    assert PF.fragment_rows(data, 100) == [r for r in range(0, 400, 3) if r >= 100]
    assert PF.fragment_rows(data, 0, [PF.RowLimitFilter(7)]) == list(range(0, 400, 3))[:7]


def limit_rule(rows, slots_of, holds, start, col, limit):
    """The closed form fbk_rows implements (include/fbk.h): the limit filter spends one row on every row in which the
    scan looks at a container; a row all of whose containers lie below the column's slot is not looked at when it
    directly follows (id + 1) a candidate row with a container in that slot or a later one."""
    c = col >> 16
    out, counted, prev = [], 0, None
    for r in (r for r in rows if r >= start):
        tail = any(s >= c for s in slots_of[r])
        if tail or not (prev is not None and prev[0] + 1 == r and prev[1]):
            if counted >= limit:
                break
            counted += 1
        if r in holds:
            out.append(r)
        prev = (r, tail)
    return out


def test_limit_filter_counts_rows_it_is_asked_about():
    """A property of the reference worth pinning (BitmapRowFilterMultiFilter.ConsiderKey, filter.go:603-622,
    consults EVERY undecided filter for a key, so BitmapRowLimitFilter :481-498 spends one of its rows on each
    row in which the scan looks at a container, whether or not the other filters go on to match it): with
    executeRowsShard's composition [column filter, limit filter] (executor.go:4139-4155) the result is "the rows
    among the first `limit` COUNTED rows that hold the column" — NOT "the first `limit` rows that hold the column";
    and the column filter's skip to (next row, column's slot) hides a following row whose containers all lie in
    lower slots.  fbk_rows reproduces exactly this (limit_rule above is its closed form); sparse and dense row ids."""
    rng = np.random.default_rng(11)
    for trial in range(200):
        span = int(rng.choice([40, 500]))
        rows = sorted(rng.choice(span, size=int(rng.integers(1, min(span, 120))), replace=False).tolist())
        data, holds, slots_of = {}, set(), {}
        for r in rows:
            slots_of[r] = [int(s) for s in rng.choice(16, size=int(rng.integers(1, 4)), replace=False)]
            for s in slots_of[r]:
                data[r * 16 + s] = PF.SetContainer(rng.integers(0, 8, 4))
        col = (int(rng.integers(0, 16)) << 16) + int(rng.integers(0, 8))
        for r in rows:
            c = data.get(r * 16 + (col >> 16))
            if c is not None and c.contains(col & 0xFFFF):
                holds.add(r)
        start = int(rng.integers(0, span))
        for limit in (1, 2, 5, 1000):
            exp = limit_rule(rows, slots_of, holds, start, col, limit)
            assert PF.fragment_rows(data, start, [PF.ColumnFilter(col), PF.RowLimitFilter(limit)]) == exp, (trial, start, col, limit)
    # rows far apart: every non-empty row is looked at, the rule is "among the first `limit` non-empty rows"
    data = {r * 16 + int(s): PF.SetContainer([1]) for r in range(0, 300, 2) for s in (r % 16, (r + 5) % 16)}
    col = (3 << 16) + 1
    holds = [r for r in range(0, 300, 2) if 3 in (r % 16, (r + 5) % 16)]
    assert PF.fragment_rows(data, 10, [PF.ColumnFilter(col), PF.RowLimitFilter(20)]) == [r for r in range(10, 50, 2) if r in holds]
    # the same rule with a rows filter in front: row 51 (not in the set, but the first key of the scan) costs one
    data = {r * 16 + (r % 16): PF.SetContainer([3]) for r in range(0, 400, 3)}
    assert PF.fragment_rows(data, 50, [PF.RowsFilter([60, 61, 63, 66, 300]), PF.RowLimitFilter(2)]) == [60]
