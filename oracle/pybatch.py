"""ctypes front end of the oracle's batch entry points (oracle/batch_oracle.c).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's checker / cpu_baseline legs).  The batch
functions run the restated reference calls of roaring_oracle.c / bsi_oracle.c over every shard of a
BASELINE.json configuration on host threads, from the same flattened descriptors the C ABI uploads —
no Python object per container, so whole configurations are checked in seconds.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import pyoracle as O

EQ, NEQ, LT, LTE, GT, GTE = 1, 2, 3, 4, 5, 6
BETWEEN = 0
OP_AND, OP_OR, OP_XOR, OP_ANDNOT = 0, 1, 2, 3

_sigs_done = False


def threads() -> int:
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def _lib() -> C.CDLL:
    global _sigs_done
    L = O.lib()
    if not _sigs_done:
        vp, i32, u32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_int64
        sig = {
            "orc_rowset_from_flat": (vp, [vp, u64, u32, vp, u64, i32]),
            "orc_rowset_from_dense": (vp, [vp, u32, i32]),
            "orc_rowset_free": (None, [vp]),
            "orc_rowset_rows": (u32, [vp]),
            "orc_rowset_row_words": (None, [vp, u32, vp]),
            "orc_rowset_words": (None, [vp, vp, i32]),
            "orc_rowset_counts": (None, [vp, vp, u64, vp]),
            "orc_batch_intersection_count": (None, [vp, vp, vp, vp, u64, vp, i32]),
            "orc_batch_intersection_count_repeat": (None, [vp, vp, vp, vp, u64, vp, i32, u64]),
            "orc_batch_setop": (vp, [i32, vp, vp, vp, vp, u64, vp, i32]),
            "orc_batch_union_n_icount": (None, [vp, vp, u64, u32, vp, vp, vp, vp, i32]),
            "orc_batch_union_n": (vp, [vp, vp, u64, u32, vp, i32]),
            "orc_batch_count_matrix": (None, [vp, vp, u32, vp, vp, u32, vp, vp, u64, vp, i32]),
            "orc_batch_topk_counts": (None, [vp, vp, u32, vp, vp, u64, vp, i32]),
            "orc_batch_bsi_range": (vp, [vp, vp, u64, u32, i32, i64, i64, vp, i32]),
            "orc_batch_bsi_sum": (None, [vp, vp, u64, u32, vp, vp, vp, vp, i32]),
            "orc_batch_bsi_minmax": (None, [vp, vp, u64, u32, i32, vp, vp, vp, vp, i32]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _sigs_done = True
    return L


def _u32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.uint32)


class RowSet:
    """Oracle Bitmaps of a set of rows (keys 0..15), addressed by row ordinal like an fbk batch."""

    def __init__(self, handle: int):
        if not handle:
            raise ValueError("oracle: malformed descriptor table")
        self.h = handle

    @classmethod
    def from_flat(cls, descs: np.ndarray, payload: np.ndarray, n_rows: int) -> "RowSet":
        """descs: the fbk_container_desc table (tests/datagen.DESC_DTYPE), payload: uint8"""
        d = np.ascontiguousarray(descs)
        assert d.dtype.itemsize == 32
        p = np.ascontiguousarray(payload.view(np.uint8).reshape(-1))
        return cls(_lib().orc_rowset_from_flat(d.ctypes.data, d.size, n_rows, p.ctypes.data, p.size, threads()))

    @classmethod
    def from_dense(cls, words: np.ndarray) -> "RowSet":
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 16 * 1024)
        return cls(_lib().orc_rowset_from_dense(w.ctypes.data, w.shape[0], threads()))

    @property
    def n_rows(self) -> int:
        return int(_lib().orc_rowset_rows(self.h))

    def words(self, nthreads: int = 0) -> np.ndarray:
        out = np.empty((self.n_rows, 16, 1024), dtype=np.uint64)
        _lib().orc_rowset_words(self.h, out.ctypes.data, nthreads or threads())
        return out

    def counts(self, rows=None) -> np.ndarray:
        r = _u32(np.arange(self.n_rows) if rows is None else rows)
        out = np.zeros(r.size, dtype=np.uint64)
        _lib().orc_rowset_counts(self.h, r.ctypes.data, r.size, out.ctypes.data)
        return out

    def free(self) -> None:
        if self.h:
            _lib().orc_rowset_free(self.h)
            self.h = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def intersection_count(A: RowSet, ra, B: RowSet, rb, nthreads: int = 0) -> np.ndarray:
    ra, rb = _u32(ra), _u32(rb)
    out = np.zeros(ra.size, dtype=np.uint64)
    _lib().orc_batch_intersection_count(A.h, ra.ctypes.data, B.h, rb.ctypes.data, ra.size, out.ctypes.data, nthreads or threads())
    return out


def intersection_count_repeat(A: RowSet, ra, B: RowSet, rb, passes: int, nthreads: int = 0) -> np.ndarray:
    """`passes` passes over the pairs on one pool of threads (for timing: thread start-up amortised)"""
    ra, rb = _u32(ra), _u32(rb)
    out = np.zeros(ra.size, dtype=np.uint64)
    _lib().orc_batch_intersection_count_repeat(A.h, ra.ctypes.data, B.h, rb.ctypes.data, ra.size, out.ctypes.data, nthreads or threads(), passes)
    return out


def setop(op: int, A: RowSet, ra, B: RowSet, rb, nthreads: int = 0):
    ra, rb = _u32(ra), _u32(rb)
    cnt = np.zeros(ra.size, dtype=np.uint64)
    h = _lib().orc_batch_setop(op, A.h, ra.ctypes.data, B.h, rb.ctypes.data, ra.size, cnt.ctypes.data, nthreads or threads())
    return RowSet(h), cnt


def union_n_intersection_count(A: RowSet, groups: np.ndarray, F: RowSet, frows, nthreads: int = 0):
    """(|∪ group ∩ F|, |∪ group|) per group"""
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    fr = _u32(frows)
    out, ucnt = np.zeros(g.shape[0], dtype=np.uint64), np.zeros(g.shape[0], dtype=np.uint64)
    _lib().orc_batch_union_n_icount(A.h, g.ctypes.data, g.shape[0], g.shape[1], F.h, fr.ctypes.data, out.ctypes.data, ucnt.ctypes.data, nthreads or threads())
    return out, ucnt


def union_n(A: RowSet, groups: np.ndarray, nthreads: int = 0):
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    ucnt = np.zeros(g.shape[0], dtype=np.uint64)
    h = _lib().orc_batch_union_n(A.h, g.ctypes.data, g.shape[0], g.shape[1], ucnt.ctypes.data, nthreads or threads())
    return RowSet(h), ucnt


def count_matrix(A: RowSet, ra: np.ndarray, B: RowSet, rb: np.ndarray, F: Optional[RowSet] = None, frows=None, nthreads: int = 0) -> np.ndarray:
    """per-shard matrices [n_shards, na, nb]: |(A_i ∩ F) ∩ B_j|"""
    ra, rb = np.ascontiguousarray(ra, dtype=np.uint32), np.ascontiguousarray(rb, dtype=np.uint32)
    ns, na, nb = ra.shape[0], ra.shape[1], rb.shape[1]
    fr = _u32(frows) if F is not None else None
    out = np.zeros((ns, na, nb), dtype=np.uint64)
    _lib().orc_batch_count_matrix(A.h, ra.ctypes.data, na, B.h, rb.ctypes.data, nb, F.h if F is not None else None,
                                  fr.ctypes.data if fr is not None else None, ns, out.ctypes.data, nthreads or threads())
    return out


def topk_counts(A: RowSet, ra: np.ndarray, F: Optional[RowSet] = None, frows=None, nthreads: int = 0) -> np.ndarray:
    ra = np.ascontiguousarray(ra, dtype=np.uint32)
    fr = _u32(frows) if F is not None else None
    out = np.zeros(ra.shape, dtype=np.uint64)
    _lib().orc_batch_topk_counts(A.h, ra.ctypes.data, ra.shape[1], F.h if F is not None else None, fr.ctypes.data if fr is not None else None,
                                 ra.shape[0], out.ctypes.data, nthreads or threads())
    return out


def bsi_range(A: RowSet, base, depth: int, op: int, predicate: int, predicate2: int = 0, nthreads: int = 0):
    b = _u32(base)
    cnt = np.zeros(b.size, dtype=np.uint64)
    h = _lib().orc_batch_bsi_range(A.h, b.ctypes.data, b.size, depth, op, predicate, predicate2, cnt.ctypes.data, nthreads or threads())
    return RowSet(h), cnt


def bsi_sum(A: RowSet, base, depth: int, F: Optional[RowSet] = None, frows=None, nthreads: int = 0):
    b = _u32(base)
    fr = _u32(frows) if F is not None else None
    s, c = np.zeros(b.size, dtype=np.int64), np.zeros(b.size, dtype=np.uint64)
    _lib().orc_batch_bsi_sum(A.h, b.ctypes.data, b.size, depth, F.h if F is not None else None, fr.ctypes.data if fr is not None else None,
                             s.ctypes.data, c.ctypes.data, nthreads or threads())
    return s, c


def bsi_minmax(A: RowSet, base, depth: int, is_max: bool, F: Optional[RowSet] = None, frows=None, nthreads: int = 0):
    b = _u32(base)
    fr = _u32(frows) if F is not None else None
    v, c = np.zeros(b.size, dtype=np.int64), np.zeros(b.size, dtype=np.uint64)
    _lib().orc_batch_bsi_minmax(A.h, b.ctypes.data, b.size, depth, int(is_max), F.h if F is not None else None,
                                fr.ctypes.data if fr is not None else None, v.ctypes.data, c.ctypes.data, nthreads or threads())
    return v, c
