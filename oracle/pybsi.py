"""ctypes front end of oracle/bsi_oracle.c (BSI Sum/Range, TopK row counts, GroupBy count
matrix, UnionRows).  TEST INFRASTRUCTURE ONLY — see oracle/pyoracle.py."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import pyoracle as O

EQ, NEQ, LT, LTE, GT, GTE = 1, 2, 3, 4, 5, 6
OPS = {"EQ": EQ, "NEQ": NEQ, "LT": LT, "LTE": LTE, "GT": GT, "GTE": GTE}

_ready = False


def _lib():
    global _ready
    L = O.lib()
    if not _ready:
        vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
        sig = {
            "orc_bsi_sum": (None, [vp, i32, vp, i32, C.POINTER(i64), C.POINTER(u64)]),
            "orc_bsi_min": (None, [vp, i32, vp, i32, u64, C.POINTER(i64), C.POINTER(u64)]),
            "orc_bsi_max": (None, [vp, i32, vp, i32, u64, C.POINTER(i64), C.POINTER(u64)]),
            "orc_bsi_range": (vp, [vp, i32, i32, u64, i64]),
            "orc_bsi_range_between": (vp, [vp, i32, u64, i64, i64]),
            "orc_bsi_range_lt_unsigned": (vp, [vp, i32, vp, u64, u64, i32]),
            "orc_bsi_range_gt_unsigned": (vp, [vp, i32, vp, u64, u64, i32]),
            "orc_bsi_range_between_unsigned": (vp, [vp, i32, vp, u64, u64, u64]),
            "orc_topk_row_counts": (None, [vp, i32, vp, i32, vp]),
            "orc_groupby_counts": (None, [vp, i32, vp, i32, vp, i32, vp]),
            "orc_union_rows": (vp, [vp, i32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _ready = True
    return L


class Fragment:
    """rows[r] = OBitmap (container keys 0..15) or None."""

    def __init__(self, rows: Sequence[Optional[O.OBitmap]]):
        self.rows = list(rows)
        self.arr = (C.c_void_p * max(len(self.rows), 1))(*[(r.p if r is not None else None) for r in self.rows])

    def __len__(self):
        return len(self.rows)


def row_from_columns(cols: Iterable[int]) -> O.OBitmap:
    """NewRow(cols...) restricted to one shard: column -> (slot = col >> 16, low bits)."""
    return O.bitmap_from_values([c & 0xFFFFF for c in cols])


def columns(bm: O.OBitmap) -> List[int]:
    return bm.slice()


def bsi_fragment_from_values(values: Dict[int, int], bit_depth: int, optimize: bool = True) -> Fragment:
    """setValue for every (column, value): exists bit, sign bit, magnitude bits
    (fragment.positionsForValue, fragment.go:619-657)."""
    rows: List[List[int]] = [[] for _ in range(bit_depth + 2)]
    for col, v in values.items():
        rows[0].append(col)
        if v < 0:
            rows[1].append(col)
        u = -v if v < 0 else v
        for i in range(bit_depth):
            if (u >> i) & 1:
                rows[2 + i].append(col)
    return Fragment([O.bitmap_from_values(r, optimize) if r else None for r in rows])


def bsi_sum(frag: Fragment, filt: Optional[O.OBitmap], has_filter: bool):
    s, c = C.c_int64(), C.c_uint64()
    _lib().orc_bsi_sum(frag.arr, len(frag), filt.p if filt is not None else None, 1 if has_filter else 0, C.byref(s), C.byref(c))
    return s.value, c.value


def bsi_min(frag: Fragment, filt: Optional[O.OBitmap], bit_depth: int):
    """fragment.min (fragment.go:754): (min, count); filt None = no filter."""
    v, c = C.c_int64(), C.c_uint64()
    _lib().orc_bsi_min(frag.arr, len(frag), filt.p if filt is not None else None, 1 if filt is not None else 0, bit_depth, C.byref(v), C.byref(c))
    return v.value, c.value


def bsi_max(frag: Fragment, filt: Optional[O.OBitmap], bit_depth: int):
    """fragment.max (fragment.go:803): (max, count); filt None = no filter."""
    v, c = C.c_int64(), C.c_uint64()
    _lib().orc_bsi_max(frag.arr, len(frag), filt.p if filt is not None else None, 1 if filt is not None else 0, bit_depth, C.byref(v), C.byref(c))
    return v.value, c.value


def bsi_range(frag: Fragment, op: int, bit_depth: int, predicate: int) -> O.OBitmap:
    return O.OBitmap(_lib().orc_bsi_range(frag.arr, len(frag), op, bit_depth, predicate))


def bsi_range_between(frag: Fragment, bit_depth: int, lo: int, hi: int) -> O.OBitmap:
    return O.OBitmap(_lib().orc_bsi_range_between(frag.arr, len(frag), bit_depth, lo, hi))


def bsi_range_lt_unsigned(frag, filt, bit_depth, pred, allow_eq):
    return O.OBitmap(_lib().orc_bsi_range_lt_unsigned(frag.arr, len(frag), filt.p, bit_depth, pred, 1 if allow_eq else 0))


def bsi_range_gt_unsigned(frag, filt, bit_depth, pred, allow_eq):
    return O.OBitmap(_lib().orc_bsi_range_gt_unsigned(frag.arr, len(frag), filt.p, bit_depth, pred, 1 if allow_eq else 0))


def bsi_range_between_unsigned(frag, filt, bit_depth, lo, hi):
    return O.OBitmap(_lib().orc_bsi_range_between_unsigned(frag.arr, len(frag), filt.p, bit_depth, lo, hi))


def topk_row_counts(frag: Fragment, filt: Optional[O.OBitmap]) -> np.ndarray:
    out = np.zeros(len(frag), dtype=np.uint64)
    _lib().orc_topk_row_counts(frag.arr, len(frag), filt.p if filt is not None else None, 1 if filt is not None else 0, out.ctypes.data)
    return out


def groupby_counts(a: Fragment, b: Fragment, filt: Optional[O.OBitmap]) -> np.ndarray:
    out = np.zeros((len(a), len(b)), dtype=np.uint64)
    _lib().orc_groupby_counts(a.arr, len(a), b.arr, len(b), filt.p if filt is not None else None, 1 if filt is not None else 0, out.ctypes.data)
    return out


def union_rows(frag: Fragment) -> O.OBitmap:
    return O.OBitmap(_lib().orc_union_rows(frag.arr, len(frag)))
