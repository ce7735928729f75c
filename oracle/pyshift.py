"""ORACLE (test infrastructure only — never imported by the product): restatement of the
reference's Shift — shift() / shiftArray / shiftBitmap / shiftRun (roaring/roaring.go:6184-6257),
Bitmap.Shift(1) (:1629-1662) and Row.Shift / RowSegment.Shift (row.go:374-396, 613-626) — on the
containers of oracle/pyoracle.py.  Pinned to TestBitmap_Shift (roaring/roaring_test.go:1389-1417)
and TestExecutor_Execute_Shift (executor_test.go:6590-6676) in tests/test_oracle_shift.py."""
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import pyoracle as O

MAX_CONTAINER_KEY = (1 << 48) - 1  # roaring.go: maxContainerKey
SHARD_WIDTH = 1 << 20


def shift_container(c: Optional[O.OContainer]) -> Tuple[Optional[O.OContainer], bool]:
    """shift(): (shifted container or None, carry out of value 65535)."""
    if c is None or c.p is None or c.n == 0:  # roaring.go:6185-6187
        return None, False
    if c.typ == O.ARRAY:  # shiftArray :6197-6210: v+1 == 0 in uint16 is the carry, dropped from the output
        a = c.data().astype(np.uint32)
        carry = bool(a.size and a[-1] == 0xFFFF)
        return O.OContainer.array((a[a != 0xFFFF] + 1).astype(np.uint16)), carry
    if c.typ == O.RUN:  # shiftRun :6230-6257
        out: List[Tuple[int, int]] = []
        carry = False
        for start, last in c.data().tolist():
            if start == 0xFFFF:  # the final run was the single bit 65535
                carry = True
                break
            if last == 0xFFFF:  # the final run ends on the container edge
                out.append((start + 1, last))
                carry = True
            else:
                out.append((start + 1, last + 1))
                carry = False
        return O.OContainer.run(out), carry
    w = c.words()  # shiftBitmap :6213-6227
    carry = bool(w[1023] >> np.uint64(63))
    out_w = (w << np.uint64(1)) | np.concatenate([np.zeros(1, np.uint64), w[:-1] >> np.uint64(63)])
    return O.OContainer.bitmap(out_w, c.n - int(carry)), carry


def _add_zero(c: Optional[O.OContainer]) -> O.OContainer:
    """o.add(0): the carried bit becomes value 0 of the next container."""
    if c is None or c.p is None or c.n == 0:
        return O.OContainer.array([0])
    w = c.words()
    w[0] |= np.uint64(1)
    return O.optimize(O.OContainer.bitmap(w))


def bitmap_shift(items: List[Tuple[int, O.OContainer]]) -> List[Tuple[int, O.OContainer]]:
    """Bitmap.Shift(1) over (key, container) pairs in key order (roaring.go:1629-1662)."""
    out: Dict[int, O.OContainer] = {}
    last_carry, last_key = False, 0
    for ki, ci in sorted(items, key=lambda kv: kv[0]):
        if last_carry and ki > last_key + 1:  # :1641-1645
            out[last_key + 1] = O.OContainer.array([0])
            last_carry = False
        o, carry = shift_container(ci)
        if last_carry:
            o = _add_zero(o)
        if o is not None and o.n > 0:
            out[ki] = o
        last_carry, last_key = carry, ki
    if last_carry and last_key != MAX_CONTAINER_KEY:  # :1655-1658
        out[last_key + 1] = O.OContainer.array([0])
    return sorted(out.items())


def row_shift(segments: Dict[int, List[Tuple[int, O.OContainer]]], n: int) -> Dict[int, List[Tuple[int, O.OContainer]]]:
    """Row.Shift(n): every segment shifted on its own, n times (row.go:374-396); a bit carried
    out of a segment stays in that segment's data under a key of the next shard."""
    work = segments
    for _ in range(n):
        work = {shard: bitmap_shift(items) for shard, items in work.items()}
    return work


def row_columns(segments: Dict[int, List[Tuple[int, O.OContainer]]]) -> List[int]:
    """Row.Columns(): the bits of every segment's data, keys are absolute (shard*16 + i)."""
    cols = set()
    for items in segments.values():
        for key, c in items:
            cols.update((key << 16) + v for v in c.values())
    return sorted(cols)


def row_from_columns(cols) -> Dict[int, List[Tuple[int, O.OContainer]]]:
    segs: Dict[int, Dict[int, List[int]]] = {}
    for c in cols:
        segs.setdefault(c // SHARD_WIDTH, {}).setdefault(c >> 16, []).append(c & 0xFFFF)
    return {s: [(k, O.optimize(O.OContainer.array(sorted(v)))) for k, v in sorted(d.items())] for s, d in segs.items()}
