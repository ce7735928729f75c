"""ctypes front end of the CPU ORACLE (oracle/libroaring_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  featurebase_amd/ never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libroaring_oracle.so")

NIL, ARRAY, BITMAP, RUN = 0, 1, 2, 3
TYPE_NAMES = {NIL: "nil", ARRAY: "array", BITMAP: "bitmap", RUN: "run"}

_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("roaring_oracle.c", "bsi_oracle.c", "wire_oracle.c", "batch_oracle.c", "roaring_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libroaring_oracle.so"])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
        sig = {
            "orc_new_array": (vp, [vp, i32]),
            "orc_new_bitmap": (vp, [vp, i32]),
            "orc_new_run": (vp, [vp, i32]),
            "orc_clone": (vp, [vp]),
            "orc_free": (None, [vp]),
            "orc_n": (i32, [vp]),
            "orc_typ": (i32, [vp]),
            "orc_len": (i32, [vp]),
            "orc_data": (vp, [vp]),
            "orc_to_words": (None, [vp, vp]),
            "orc_count": (i32, [vp]),
            "orc_count_range": (i32, [vp, i32, i32]),
            "orc_array_count_range": (i32, [vp, i32, i32, i32]),
            "orc_words_count_range": (i32, [vp, i32, i32]),
            "orc_run_count_range": (i32, [vp, i32, i32, i32]),
            "orc_count_runs": (i32, [vp]),
            "orc_optimize": (vp, [vp]),
            "orc_array_to_bitmap": (vp, [vp]),
            "orc_bitmap_to_array": (vp, [vp]),
            "orc_run_to_bitmap": (vp, [vp]),
            "orc_bitmap_to_run": (vp, [vp]),
            "orc_array_to_run": (vp, [vp]),
            "orc_run_to_array": (vp, [vp]),
            "orc_bitmap_set_range": (None, [vp, u64, u64]),
            "orc_bitmap_xor_range": (None, [vp, u64, u64]),
            "orc_bitmap_zero_range": (None, [vp, u64, u64]),
            "orc_intersection_count": (i32, [vp, vp]),
            "orc_intersect": (vp, [vp, vp]),
            "orc_union": (vp, [vp, vp]),
            "orc_difference": (vp, [vp, vp]),
            "orc_xor": (vp, [vp, vp]),
            "orc_union_in_place": (vp, [vp, vp]),
            "orc_bitwise_compare": (i32, [vp, vp]),
            "orc_bitmap_new": (vp, []),
            "orc_bitmap_free": (None, [vp]),
            "orc_bitmap_put": (None, [vp, u64, vp]),
            "orc_bitmap_len": (i32, [vp]),
            "orc_bitmap_key": (u64, [vp, i32]),
            "orc_bitmap_container": (vp, [vp, i32]),
            "orc_bitmap_count": (u64, [vp]),
            "orc_bitmap_count_range": (u64, [vp, u64, u64]),
            "orc_bitmap_intersection_count": (u64, [vp, vp]),
            "orc_bitmap_intersect": (vp, [vp, vp]),
            "orc_bitmap_union": (vp, [vp, vp, i32]),
            "orc_bitmap_difference": (vp, [vp, vp, i32]),
            "orc_bitmap_xor": (vp, [vp, vp]),
            "orc_set_n": (None, [vp, i32]),
            "orc_flip": (vp, [vp]),
            "orc_run_append_interval": (i32, [vp, i32, C.c_uint32]),
            "orc_kernel": (vp, [C.c_char_p, vp, vp]),
            "orc_count_kernel": (i32, [C.c_char_p, vp, vp]),
            "orc_dense_intersection_count": (u64, [vp, vp, u64, vp]),
            "orc_dense_intersect_count": (u64, [vp, vp, u64, vp, vp]),
            "orc_roaring_marshal": (vp, [vp, i32, C.POINTER(u64)]),
            "orc_wire_free": (None, [vp]),
            "orc_roaring_unmarshal": (vp, [C.c_char_p, u64, C.POINTER(i32)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class OContainer:
    """Owning handle of an orc_container* (None handle = nil container)."""

    def __init__(self, ptr: Optional[int], own: bool = True):
        self.p = C.c_void_p(ptr) if ptr else None
        self.own = own

    def __del__(self):
        if getattr(self, "own", False) and self.p and _lib is not None:
            _lib.orc_free(self.p)
            self.p = None

    # -- constructors -------------------------------------------------------------
    @staticmethod
    def array(values: Iterable[int]) -> "OContainer":
        a = np.ascontiguousarray(np.asarray(list(values) if not isinstance(values, np.ndarray) else values, dtype=np.uint16))
        return OContainer(lib().orc_new_array(a.ctypes.data, a.size))

    @staticmethod
    def bitmap(words, n: int = -1) -> "OContainer":
        w = np.zeros(1024, dtype=np.uint64)
        src = np.asarray(words, dtype=np.uint64)
        w[: src.size] = src
        return OContainer(lib().orc_new_bitmap(w.ctypes.data, n))

    @staticmethod
    def run(intervals: Iterable[Tuple[int, int]]) -> "OContainer":
        r = np.ascontiguousarray(np.asarray(list(intervals), dtype=np.uint16).reshape(-1, 2))
        return OContainer(lib().orc_new_run(r.ctypes.data, r.shape[0]))

    @staticmethod
    def from_words(words: np.ndarray, typ: int) -> Optional["OContainer"]:
        """Build a container of the requested encoding from 1024 words of bit content."""
        bm = OContainer.bitmap(words)
        if typ == BITMAP:
            return bm
        if typ == ARRAY:
            return OContainer(lib().orc_bitmap_to_array(bm.p))
        return OContainer(lib().orc_bitmap_to_run(bm.p))

    # -- accessors ------------------------------------------------------------------
    @property
    def typ(self) -> int:
        return lib().orc_typ(self.p)

    @property
    def n(self) -> int:
        return lib().orc_n(self.p)

    @property
    def length(self) -> int:
        return lib().orc_len(self.p)

    def data(self) -> np.ndarray:
        t, ln = self.typ, self.length
        if t == NIL:
            return np.zeros(0, dtype=np.uint16)
        addr = lib().orc_data(self.p)
        if t == ARRAY:
            return np.ctypeslib.as_array((C.c_uint16 * ln).from_address(addr)).copy() if ln else np.zeros(0, np.uint16)
        if t == RUN:
            return (
                np.ctypeslib.as_array((C.c_uint16 * (2 * ln)).from_address(addr)).copy().reshape(-1, 2)
                if ln
                else np.zeros((0, 2), np.uint16)
            )
        return np.ctypeslib.as_array((C.c_uint64 * 1024).from_address(addr)).copy()

    def words(self) -> np.ndarray:
        out = np.zeros(1024, dtype=np.uint64)
        lib().orc_to_words(self.p, out.ctypes.data)
        return out

    def values(self) -> List[int]:
        bits = np.unpackbits(self.words().view(np.uint8), bitorder="little")
        return np.nonzero(bits)[0].tolist()

    def clone(self) -> "OContainer":
        return OContainer(lib().orc_clone(self.p))

    def __repr__(self):
        return f"OContainer({TYPE_NAMES[self.typ]}, n={self.n}, len={self.length})"


def _wrap(ptr) -> OContainer:
    return OContainer(ptr)


def intersection_count(a: OContainer, b: OContainer) -> int:
    return lib().orc_intersection_count(a.p, b.p)


def intersect(a, b):
    return _wrap(lib().orc_intersect(a.p, b.p))


def union(a, b):
    return _wrap(lib().orc_union(a.p, b.p))


def difference(a, b):
    return _wrap(lib().orc_difference(a.p, b.p))


def xor(a, b):
    return _wrap(lib().orc_xor(a.p, b.p))


def union_in_place(a, b):
    return _wrap(lib().orc_union_in_place(a.p, b.p))


def optimize(a):
    return _wrap(lib().orc_optimize(a.p))


def bitwise_compare(a, b) -> int:
    return lib().orc_bitwise_compare(a.p, b.p)


def flip(a):
    return _wrap(lib().orc_flip(a.p))


# NOTE: always pass OContainer objects (not `.p` of a temporary): a temporary is freed as
# soon as its `.p` has been read, before the C call runs.
def array_to_bitmap(a):
    return _wrap(lib().orc_array_to_bitmap(a.p))


def bitmap_to_array(a):
    return _wrap(lib().orc_bitmap_to_array(a.p))


def run_to_bitmap(a):
    return _wrap(lib().orc_run_to_bitmap(a.p))


def bitmap_to_run(a):
    return _wrap(lib().orc_bitmap_to_run(a.p))


def array_to_run(a):
    return _wrap(lib().orc_array_to_run(a.p))


def run_to_array(a):
    return _wrap(lib().orc_run_to_array(a.p))


def count_runs(a) -> int:
    return lib().orc_count_runs(a.p)


def count(a) -> int:
    return lib().orc_count(a.p)


def runs_of_content(a):
    """Maximal runs of a container's bit content, as [(start, last)]."""
    bm = OContainer.bitmap(a.words())
    r = bitmap_to_run(bm)
    return [tuple(x) for x in r.data().tolist()]


def kernel(name: str, a, b):
    """Call one Go type-pair kernel by its reference name (e.g. 'intersectRunRun')."""
    return _wrap(lib().orc_kernel(name.encode(), a.p, b.p))


def count_kernel(name: str, a, b) -> int:
    return lib().orc_count_kernel(name.encode(), a.p, b.p)


def run_append_interval(base, iv) -> int:
    r = np.ascontiguousarray(np.asarray(list(base), dtype=np.uint16).reshape(-1, 2))
    packed = (int(iv[0]) & 0xFFFF) | ((int(iv[1]) & 0xFFFF) << 16)  # struct {u16,u16} by value
    return lib().orc_run_append_interval(r.ctypes.data, r.shape[0], packed)


OPS = {"intersect": intersect, "union": union, "difference": difference, "xor": xor}


class OBitmap:
    """Owning handle of an orc_bitmap* (sorted keys + containers)."""

    def __init__(self, ptr: Optional[int] = None):
        self.p = C.c_void_p(ptr if ptr else lib().orc_bitmap_new())

    def __del__(self):
        if self.p and _lib is not None:
            _lib.orc_bitmap_free(self.p)
            self.p = None

    @staticmethod
    def from_containers(items: Sequence[Tuple[int, Optional[OContainer]]]) -> "OBitmap":
        b = OBitmap()
        for key, c in sorted(items, key=lambda kv: kv[0]):
            # the bitmap takes ownership: hand it a clone
            b.put(key, c)
        return b

    def put(self, key: int, c: Optional[OContainer]) -> None:
        lib().orc_bitmap_put(self.p, key, lib().orc_clone(c.p) if c is not None and c.p else None)

    def __len__(self):
        return lib().orc_bitmap_len(self.p)

    def items(self) -> List[Tuple[int, OContainer]]:
        out = []
        for i in range(len(self)):
            k = lib().orc_bitmap_key(self.p, i)
            c = lib().orc_bitmap_container(self.p, i)
            out.append((k, OContainer(lib().orc_clone(c)) if c else OContainer(None)))
        return out

    def count(self) -> int:
        return lib().orc_bitmap_count(self.p)

    def count_range(self, s: int, e: int) -> int:
        return lib().orc_bitmap_count_range(self.p, s, e)

    def intersection_count(self, other: "OBitmap") -> int:
        return lib().orc_bitmap_intersection_count(self.p, other.p)

    def intersect(self, other: "OBitmap") -> "OBitmap":
        return OBitmap(lib().orc_bitmap_intersect(self.p, other.p))

    def xor(self, other: "OBitmap") -> "OBitmap":
        return OBitmap(lib().orc_bitmap_xor(self.p, other.p))

    def _multi(self, fn, others: Sequence["OBitmap"]) -> "OBitmap":
        arr = (C.c_void_p * len(others))(*[o.p for o in others])
        return OBitmap(fn(self.p, arr, len(others)))

    def union(self, *others: "OBitmap") -> "OBitmap":
        return self._multi(lib().orc_bitmap_union, others)

    def difference(self, *others: "OBitmap") -> "OBitmap":
        return self._multi(lib().orc_bitmap_difference, others)

    def marshal(self, optimize_first: bool = True) -> bytes:
        """Bitmap.WriteTo (roaring.go:1730): Pilosa roaring format."""
        n = C.c_uint64()
        p = lib().orc_roaring_marshal(self.p, 1 if optimize_first else 0, C.byref(n))
        try:
            return C.string_at(p, n.value)
        finally:
            lib().orc_wire_free(p)

    @staticmethod
    def unmarshal(data: bytes) -> "OBitmap":
        """Bitmap.UnmarshalBinary (Pilosa or official roaring format); ValueError on the
        conditions the reference's iterators report as errors."""
        err = C.c_int32()
        p = lib().orc_roaring_unmarshal(data, len(data), C.byref(err))
        if err.value or not p:
            raise ValueError("malformed roaring data")
        return OBitmap(p)

    def slice(self) -> List[int]:
        """All set bit positions (Bitmap.Slice, roaring.go:623)."""
        out: List[int] = []
        for k, c in self.items():
            out.extend((k << 16) + v for v in c.values())
        return out


def bitmap_from_values(values: Iterable[int], optimize_containers: bool = True) -> OBitmap:
    """NewFileBitmap(values...) + Optimize(): group by high 48 bits, encode each container
    by Container.optimize() (roaring.go:3412)."""
    groups = {}
    for v in values:
        groups.setdefault(v >> 16, []).append(v & 0xFFFF)
    b = OBitmap()
    for key in sorted(groups):
        lo = sorted(set(groups[key]))
        c = OContainer.array(lo)
        if optimize_containers:
            c = optimize(c)
        b.put(key, c)
    return b
