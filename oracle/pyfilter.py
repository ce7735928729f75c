"""TEST INFRASTRUCTURE ONLY (oracle): Python restatement of the reference's bitmap filter protocol,
roaring/filter.go — FilterKey / FilterResult (:30-173), BitmapColumnFilter (:226-249), BitmapRowsFilter
(:252-293), BitmapRowFilterBase (:371-469), BitmapRowLimitFilter (:471-509), BitmapRowFilterSingleFilter
(:512-546), BitmapRowFilterMultiFilter (:551-681), NewBitmapRowFilter (:790-798), ApplyFilterToIterator
(:1062-1085) — and of fragment.rows (fragment.go:2465-2486) on top of it.

The product never imports this module: only tests/ do, as the checker of fbk_rows.  It is pinned by the
expectations of the reference's own tests (roaring/filter_internal_test.go TestBaseFilter, TestColumnFilter,
TestRowsFilter; fragment_internal_test.go TestFragment_RowsIteration), restated in tests/test_oracle_filter.py
— those tests build their data and their expectations with loops, not tables, so there is nothing to extract
mechanically; the loops are restated next to a citation of each.

Containers are anything with `.n` (cardinality) and `contains(v)`; the module ships `SetContainer` (a Python
set of 16-bit values) and `wrap(OContainer)` for the C oracle's containers.
"""
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

ROW_EXPONENT = 4  # shardwidth.Exponent - 16 (filter.go:24)
ROW_WIDTH = 1 << ROW_EXPONENT
KEY_MASK = ROW_WIDTH - 1
ROW_MASK = ~KEY_MASK & 0xFFFFFFFFFFFFFFFF
MAX_KEY = 0xFFFFFFFFFFFFFFFF


class FilterError(Exception):
    pass


class Result:
    """FilterResult (filter.go:41-45): exclusive upper bounds of a run of matches, then of a run of rejections."""

    __slots__ = ("yes", "no", "err")

    def __init__(self, yes: int = 0, no: int = 0, err: Optional[str] = None):
        self.yes, self.no, self.err = yes, no, err

    def __repr__(self):
        return f"Result(yes={self.yes}, no={self.no}, err={self.err})"


# ---- FilterKey methods (filter.go:48-173); a key is a plain int ----
def key_row(f: int) -> int:
    return f >> ROW_EXPONENT


def match_reject(y: int, n: int) -> Result:
    return Result(y, n)


def match_one(f: int) -> Result:
    return Result(f + 1, f + 1)


def need_data(f: int) -> Result:
    return Result()


def fail(msg: str) -> Result:
    return Result(err=msg)


def match_row(f: int) -> Result:
    return Result(yes=(f & ROW_MASK) + ROW_WIDTH)


def match_one_reject_row(f: int) -> Result:
    return Result(f + 1, (f & ROW_MASK) + ROW_WIDTH)


def reject_one(f: int) -> Result:
    return Result(no=f + 1)


def reject_row(f: int) -> Result:
    return Result(no=(f & ROW_MASK) + ROW_WIDTH)


def reject_until(f: int, until: int) -> Result:
    return Result(no=until)


def reject_until_row(f: int, row_id: int) -> Result:
    return Result(no=row_id << ROW_EXPONENT)


def match_row_until_row(f: int, row_id: int) -> Result:
    return Result((f & ROW_MASK) + ROW_WIDTH, row_id << ROW_EXPONENT)


def reject_until_offset(f: int, offset: int) -> Result:
    nxt = (f & ROW_MASK) + offset
    if nxt <= f:
        nxt += ROW_WIDTH
    return Result(no=nxt)


def match_one_until_offset(f: int, offset: int) -> Result:
    r = reject_until_offset(f, offset)
    r.yes = f + 1
    return r


def done(f: int) -> Result:
    return Result(no=MAX_KEY)


def match_row_and_done(f: int) -> Result:
    return Result((f & ROW_MASK) + ROW_WIDTH, MAX_KEY)


def match_one_until_same_offset(f: int) -> Result:
    return match_one_until_offset(f, f & KEY_MASK)


# ---- containers ----
class SetContainer:
    def __init__(self, values: Iterable[int]):
        self.v = set(int(x) for x in values)

    @property
    def n(self) -> int:
        return len(self.v)

    def contains(self, x: int) -> bool:
        return x in self.v


class _Wrapped:
    def __init__(self, oc):
        self.oc = oc
        self._w = None

    @property
    def n(self) -> int:
        return int(self.oc.n)

    def contains(self, x: int) -> bool:
        if self._w is None:
            self._w = self.oc.words()
        return bool((int(self._w[x >> 6]) >> (x & 63)) & 1)


def wrap(oc):
    """An oracle.pyoracle.OContainer as a filter-protocol container (Container.Contains, roaring.go)."""
    return _Wrapped(oc)


# ---- filters ----
class ColumnFilter:
    """BitmapColumnFilter (filter.go:226-249)."""

    def __init__(self, col: int):
        self.key = (col >> 16) & KEY_MASK
        self.offset = col & 0xFFFF

    def consider_key(self, key: int, n: int) -> Result:
        if (key & KEY_MASK) != self.key:
            return reject_until_offset(key, self.key)
        return need_data(key)

    def consider_data(self, key: int, data) -> Result:
        if data.contains(self.offset):
            return match_one_until_same_offset(key)
        return reject_until_offset(key, self.key)


class RowsFilter:
    """BitmapRowsFilter (filter.go:252-293): containers of any of a sorted list of rows."""

    def __init__(self, rows: Sequence[int]):
        self.rows = list(rows)
        self.i = 0 if self.rows else -1

    def consider_key(self, key: int, n: int) -> Result:
        if self.i == -1:
            return done(key)
        if n == 0:
            return reject_one(key)
        row = key >> ROW_EXPONENT
        while self.rows[self.i] < row:
            self.i += 1
            if self.i >= len(self.rows):
                self.i = -1
                return done(key)
        if self.rows[self.i] > row:
            return reject_until_row(key, self.rows[self.i])
        if self.i + 1 < len(self.rows):
            return match_row_until_row(key, self.rows[self.i + 1])
        return match_row_and_done(key)

    def consider_data(self, key: int, data) -> Result:
        return fail("bitmap rows filter should never consider data")


class RowFilterBase:
    """BitmapRowFilterBase (filter.go:371-469)."""

    def __init__(self, callback: Optional[Callable[[int], None]]):
        self.res = Result()
        self.callback = callback
        self.last_row = MAX_KEY

    def determine_by_key(self, key: int) -> Tuple[Result, bool]:
        b = self.res
        if b.err is not None:
            return b, True
        row = key_row(key)
        if b.yes <= key and b.no > key:
            return reject_until(key, b.no), True
        if self.last_row == row:
            return reject_row(key), True
        if b.yes > key:
            self.last_row = row
            if self.callback is not None:
                self.callback(row)
            res = match_one_reject_row(key)
            if b.no < res.no:
                b.no = res.no
            if b.yes <= res.no and b.no > res.no:
                res.no = b.no
            return res, True
        return b, False

    def set_result(self, key: int, result: Result) -> Result:
        self.res = result
        result, _ = self.determine_by_key(key)
        return result

    def consider_key(self, key: int, n: int) -> Result:
        self.res, fin = self.determine_by_key(key)
        if fin:
            return self.res
        if n == 0:
            return reject_one(key)
        self.res = match_one_reject_row(key)
        self.last_row = key_row(key)
        if self.callback is not None:
            self.callback(self.last_row)
        return self.res

    def consider_data(self, key: int, data) -> Result:
        self.res.err = "base iterator should never consider data"
        return self.res


class RowLimitFilter(RowFilterBase):
    """BitmapRowLimitFilter (filter.go:471-509)."""

    def __init__(self, limit: int):
        super().__init__(None)
        self.limit = limit

    def consider_key(self, key: int, n: int) -> Result:
        self.res, fin = self.determine_by_key(key)
        if fin:
            return self.res
        if n == 0:
            return reject_one(key)
        if self.limit > 0:
            self.res = match_row(key)
            self.limit -= 1
        else:
            self.res = done(key)
        return self.res

    def consider_data(self, key: int, data) -> Result:
        self.res.err = "limit iterator should never consider data"
        return self.res


class RowFilterSingle(RowFilterBase):
    """BitmapRowFilterSingleFilter (filter.go:512-546)."""

    def __init__(self, callback, flt):
        super().__init__(callback)
        self.filter = flt

    def consider_key(self, key: int, n: int) -> Result:
        res, fin = self.determine_by_key(key)
        if fin:
            return res
        return self.set_result(key, self.filter.consider_key(key, n))

    def consider_data(self, key: int, data) -> Result:
        self.res = self.filter.consider_data(key, data)
        if self.res.err is not None:
            return self.res
        res, fin = self.determine_by_key(key)
        if fin:
            return res
        self.res.err = "inner filter didn't make a decision"
        return self.res


class RowFilterMulti(RowFilterBase):
    """BitmapRowFilterMultiFilter (filter.go:551-681)."""

    def __init__(self, callback, filters):
        super().__init__(callback)
        self.filters = list(filters)
        self.yes_keys = [0] * len(self.filters)
        self.no_keys = [0] * len(self.filters)
        self.todo: List[int] = []

    def consider_key(self, key: int, n: int) -> Result:
        res, fin = self.determine_by_key(key)
        if fin:
            return res
        highest_no = key
        lowest_yes = MAX_KEY
        lowest_yes_no = 0
        self.todo = []
        for i, yk in enumerate(self.yes_keys):
            if yk > key:
                if yk < lowest_yes:
                    lowest_yes = yk
                    lowest_yes_no = self.no_keys[i]
                continue
            nk = self.no_keys[i]
            if nk > highest_no:
                highest_no = nk
                continue
            self.todo.append(i)
        if highest_no > key:
            return self.set_result(key, reject_until(key, highest_no))
        new_todo = []
        for f in self.todo:
            result = self.filters[f].consider_key(key, n)
            if result.err is not None:
                return fail(result.err)
            yk, nk = result.yes, result.no
            self.yes_keys[f], self.no_keys[f] = yk, nk
            if yk > key:
                if lowest_yes == 0 or yk < lowest_yes:
                    lowest_yes = yk
                    lowest_yes_no = nk
                continue
            if nk > highest_no:
                highest_no = nk
                continue
            new_todo.append(f)
        if highest_no > key:
            return self.set_result(key, reject_until(key, highest_no))
        self.todo = new_todo
        if self.todo:
            return need_data(key)
        if lowest_yes <= key:
            return fail(f"got lowest yes {lowest_yes} for key {key}, this shouldn't happen")
        return self.set_result(key, match_reject(lowest_yes, lowest_yes_no))

    def consider_data(self, key: int, data) -> Result:
        res, fin = self.determine_by_key(key)
        if fin:
            return res
        highest_no = key
        for f in self.todo:
            result = self.filters[f].consider_data(key, data)
            if result.err is not None:
                return fail(result.err)
            yk, nk = result.yes, result.no
            self.yes_keys[f], self.no_keys[f] = yk, nk
            if yk <= key and nk > highest_no:
                highest_no = nk
        if highest_no > key:
            return self.set_result(key, reject_until(key, highest_no))
        lowest_yes = MAX_KEY
        lowest_yes_no = key
        for i, yk in enumerate(self.yes_keys):
            if yk < lowest_yes:
                lowest_yes = yk
                lowest_yes_no = self.no_keys[i]
        if lowest_yes <= key:
            return fail(f"got lowest yes {lowest_yes} on data for key {key}, this shouldn't happen")
        return self.set_result(key, match_reject(lowest_yes, lowest_yes_no))


def new_row_filter(callback, *filters):
    """NewBitmapRowFilter (filter.go:790-798)."""
    if not filters:
        return RowFilterBase(callback)
    if len(filters) == 1:
        return RowFilterSingle(callback, filters[0])
    return RowFilterMulti(callback, filters)


def apply_filter_to_iterator(flt, containers: Iterable[Tuple[int, object]], start_key: int = 0, stats: Optional[dict] = None) -> None:
    """ApplyFilterToIterator (filter.go:1062-1085) over (key, container) pairs in ascending key order, from
    start_key on (Tx.ApplyFilter's ckey, rbf/tx.go:1663).  stats (optional) counts the ConsiderKey / ConsiderData
    calls and the containers skipped by the YesKey / NoKey look-ahead."""
    until = 0
    for key, data in containers:
        if key < start_key:
            continue
        if until >= MAX_KEY:
            break
        if key < until:
            if stats is not None:
                stats["skipped"] = stats.get("skipped", 0) + 1
            continue
        result = flt.consider_key(key, data.n)
        if stats is not None:
            stats["consider_key"] = stats.get("consider_key", 0) + 1
        if result.err is not None:
            raise FilterError(result.err)
        until = result.no
        if key < until:
            continue
        result = flt.consider_data(key, data)
        if stats is not None:
            stats["consider_data"] = stats.get("consider_data", 0) + 1
        if result.err is not None:
            raise FilterError(result.err)
        until = result.no


def fragment_rows(containers: Dict[int, object], start: int = 0, filters: Sequence = (), stats: Optional[dict] = None) -> List[int]:
    """fragment.rows(start, filters...) (fragment.go:2465-2486): the ids of the rows, from `start` on, that hold a
    container every filter matches.  containers: key (row * 16 + slot) -> container."""
    rows: List[int] = []
    flt = new_row_filter(rows.append, *filters)
    apply_filter_to_iterator(flt, sorted(containers.items()), start << ROW_EXPONENT, stats)
    return rows
