"""Host-side driver of the fbk C ABI, named after the reference types it stands in for.

    Container  ~ roaring.Container          (roaring/container_stash.go:46-53)
    Batch      ~ a set of fragment.row()s   (fragment.go:283-333), device resident
    Plan       ~ one (query, node) batch of per-shard map calls (executor.go:6742)
    Context    ~ one GPU

This is test/bench plumbing over ctypes; all arithmetic happens in libfbk.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import lib as L

SLOTS = 16
BITMAP_WORDS = 1024


DESC_DTYPE = np.dtype(
    [("key", "<u8"), ("off", "<u8"), ("row", "<u4"), ("len", "<u4"), ("n", "<i4"), ("type", "u1"), ("pad", "u1", (3,))]
)  # == fbk_container_desc (include/fbk.h), 32 bytes


@dataclass
class Container:
    """One roaring container in its wire encoding (roaring.go:53-58)."""

    typ: int  # L.TYPE_ARRAY / TYPE_BITMAP / TYPE_RUN
    data: np.ndarray  # array: uint16[n]; bitmap: uint64[1024]; run: uint16[runs, 2] (start, last)
    n: int = -1  # cardinality, -1 = unknown (device recounts)

    @staticmethod
    def array(values: Iterable[int]) -> "Container":
        a = np.asarray(list(values) if not isinstance(values, np.ndarray) else values, dtype=np.uint16)
        return Container(L.TYPE_ARRAY, a, int(a.size))

    @staticmethod
    def bitmap(words: np.ndarray, n: int = -1) -> "Container":
        w = np.ascontiguousarray(words, dtype=np.uint64)
        if w.size < BITMAP_WORDS:  # tests write short literals, like the reference's do
            w = np.concatenate([w, np.zeros(BITMAP_WORDS - w.size, dtype=np.uint64)])
        assert w.size == BITMAP_WORDS
        return Container(L.TYPE_BITMAP, w, n)

    @staticmethod
    def run(intervals: Iterable[Tuple[int, int]], n: int = -1) -> "Container":
        r = np.asarray(list(intervals), dtype=np.uint16).reshape(-1, 2)
        if n < 0:
            n = int((r[:, 1].astype(np.int64) - r[:, 0].astype(np.int64) + 1).sum()) if r.size else 0
        return Container(L.TYPE_RUN, r, n)

    @property
    def length(self) -> int:
        if self.typ == L.TYPE_ARRAY:
            return int(self.data.size)
        if self.typ == L.TYPE_RUN:
            return int(self.data.shape[0])
        return BITMAP_WORDS

    def payload(self) -> bytes:
        return np.ascontiguousarray(self.data).tobytes()

    def words(self) -> np.ndarray:
        """Bit content as uint64[1024] (representation independent, cf. BitwiseEqual
        roaring.go:6857-6922)."""
        if self.typ == L.TYPE_BITMAP:
            return self.data.copy()
        bits = np.zeros(65536, dtype=np.uint8)
        if self.typ == L.TYPE_ARRAY:
            bits[self.data.astype(np.int64)] = 1
        else:
            for s, l in self.data.astype(np.int64):
                bits[s : l + 1] = 1
        return np.packbits(bits, bitorder="little").view(np.uint64)


Row = Dict[int, Container]  # container key -> container; slot = key & 15


class Batch:
    def __init__(self, ctx: "Context", handle: int):
        self.ctx = ctx
        self.h = C.c_void_p(handle)
        self.owned = True  # False once handed to the device fragment cache

    def info(self) -> Tuple[int, int, int]:
        nr, nc, pb = C.c_uint32(), C.c_uint64(), C.c_uint64()
        L.check(self.ctx.lib.fbk_batch_info(self.ctx.h, self.h, C.byref(nr), C.byref(nc), C.byref(pb)))
        return nr.value, nc.value, pb.value

    @property
    def n_rows(self) -> int:
        return self.info()[0]

    def memory(self) -> Tuple[int, int, int]:
        """(arena bytes, heavy-row shadow bytes, shadow state 0 / 1 / 2) — fbk_batch_memory"""
        a, sh, st = C.c_uint64(), C.c_uint64(), C.c_int32()
        L.check(self.ctx.lib.fbk_batch_memory(self.ctx.h, self.h, C.byref(a), C.byref(sh), C.byref(st)))
        return a.value, sh.value, st.value

    def compact(self) -> int:
        """fbk_batch_compact: the containers into a right-sized arena; returns the arena bytes afterwards"""
        a = C.c_uint64()
        L.check(self.ctx.lib.fbk_batch_compact(self.ctx.h, self.h, C.byref(a)))
        return a.value

    def count(self, rows: Sequence[int]) -> np.ndarray:
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros(r.size, dtype=np.uint64)
        L.check(self.ctx.lib.fbk_count(self.ctx.h, self.h, r.ctypes.data, r.size, out.ctypes.data))
        return out

    def download_flat(self) -> Tuple[np.ndarray, np.ndarray, int]:
        """(descs, payload, n_rows): the batch in the flattened layout fbk_batch_upload takes — descs a
        structured array with the fields of fbk_container_desc, payload uint8."""
        n_rows, nc, pb = self.info()
        descs = np.zeros(max(nc, 1), dtype=DESC_DTYPE)
        payload = np.zeros(max(pb, 1), dtype=np.uint8)
        L.check(self.ctx.lib.fbk_batch_download(self.ctx.h, self.h, descs.ctypes.data_as(C.POINTER(L.ContainerDesc)), nc, payload.ctypes.data, pb))
        return descs[:nc], payload, n_rows

    def download(self) -> List[Row]:
        n_rows, nc, pb = self.info()
        descs = (L.ContainerDesc * max(nc, 1))()
        payload = np.zeros(max(pb, 1), dtype=np.uint8)
        L.check(self.ctx.lib.fbk_batch_download(self.ctx.h, self.h, descs, nc, payload.ctypes.data, pb))
        rows: List[Row] = [dict() for _ in range(n_rows)]
        for i in range(nc):
            d = descs[i]
            if d.type == L.TYPE_ARRAY:
                data = payload[d.off : d.off + 2 * d.len].view(np.uint16).copy()
            elif d.type == L.TYPE_RUN:
                data = payload[d.off : d.off + 4 * d.len].view(np.uint16).reshape(-1, 2).copy()
            else:
                data = payload[d.off : d.off + 8192].view(np.uint64).copy()
            rows[d.row][int(d.key)] = Container(int(d.type), data, int(d.n))
        return rows

    def to_roaring(self) -> bytes:
        """The batch in the Pilosa roaring format (Bitmap.WriteTo, roaring.go:1730)."""
        n = C.c_uint64()
        L.check(self.ctx.lib.fbk_batch_roaring_size(self.ctx.h, self.h, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        got = C.c_uint64()
        L.check(self.ctx.lib.fbk_batch_download_roaring(self.ctx.h, self.h, buf.ctypes.data, n.value, C.byref(got)))
        return buf[: got.value].tobytes()

    def free(self) -> None:
        if self.h and self.owned:
            L.check(self.ctx.lib.fbk_batch_free(self.ctx.h, self.h))
        self.h = C.c_void_p(None)


class Plan:
    def __init__(self, ctx: "Context", handle: int, n_pairs: int):
        self.ctx, self.h, self.n_pairs = ctx, C.c_void_p(handle), n_pairs

    def intersection_count(self) -> None:
        L.check(self.ctx.lib.fbk_plan_intersection_count(self.ctx.h, self.h))

    def intersection_count_total(self, device_ptr: int = 0) -> None:
        """Counts + per-node total in one launch (executeCount's mapFn + reduceFn)."""
        L.check(self.ctx.lib.fbk_plan_intersection_count_total(self.ctx.h, self.h, C.c_void_p(device_ptr or None)))

    def intersection_count_accumulate(self, device_ptr: int) -> None:
        """Counts + per-node reduce by accumulation into a zeroed uint64 at device_ptr (one launch)."""
        L.check(self.ctx.lib.fbk_plan_intersection_count_accumulate(self.ctx.h, self.h, C.c_void_p(device_ptr)))

    def setop(self, op: int, flags: int = 0) -> None:
        L.check(self.ctx.lib.fbk_plan_setop(self.ctx.h, self.h, op, flags))

    def total(self, device_ptr: int = 0) -> None:
        L.check(self.ctx.lib.fbk_plan_total(self.ctx.h, self.h, C.c_void_p(device_ptr or None)))

    def read(self, want_total: bool = False):
        counts = np.zeros(self.n_pairs, dtype=np.uint64)
        tot = C.c_uint64()
        L.check(
            self.ctx.lib.fbk_plan_read(
                self.ctx.h, self.h, counts.ctypes.data, C.cast(C.byref(tot), C.c_void_p) if want_total else None
            )
        )
        return (counts, tot.value) if want_total else counts

    def output(self) -> Batch:
        """Borrowed handle of the plan's set-op output (do not free)."""
        h = C.c_void_p()
        L.check(self.ctx.lib.fbk_plan_output(self.ctx.h, self.h, C.byref(h)))
        return Batch(self.ctx, h.value)

    def free(self) -> None:
        if self.h:
            L.check(self.ctx.lib.fbk_plan_free(self.ctx.h, self.h))
            self.h = C.c_void_p(None)


class Query:
    """A prepared query (fbk_query_*): row lists and result buffers resident, every run launch-only."""

    def __init__(self, ctx: "Context", handle: int, kind: str, shape: Tuple[int, ...]):
        self.ctx, self.h, self.kind, self.shape = ctx, C.c_void_p(handle), kind, shape

    def run(self, device_ptr: int = 0, accumulate: bool = False) -> None:
        L.check(self.ctx.lib.fbk_query_run(self.ctx.h, self.h, C.c_void_p(device_ptr or None), L.QUERY_ACCUMULATE if accumulate else 0))

    def result_ptr(self) -> Tuple[int, int]:
        """(device pointer, bytes) of the query's own result buffer"""
        p, n = C.c_void_p(), C.c_uint64()
        L.check(self.ctx.lib.fbk_query_result(self.ctx.h, self.h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def read(self, per_shard: bool = False):
        if self.kind == "count_matrix":
            n_shards, n_a, n_b = self.shape
            tot = np.zeros((n_a, n_b), dtype=np.uint64)
            ps = np.zeros((n_shards, n_a, n_b), dtype=np.uint64) if per_shard else None
            L.check(self.ctx.lib.fbk_query_read(self.ctx.h, self.h, tot.ctypes.data, ps.ctypes.data if ps is not None else None))
            return (tot, ps) if per_shard else tot
        if self.kind in ("fold", "rows"):  # counts per group / cardinalities of the result rows
            out = np.zeros(self.shape[0], dtype=np.uint64)
            L.check(self.ctx.lib.fbk_query_read(self.ctx.h, self.h, out.ctypes.data, None))
            return out
        if self.kind == "topn":  # (row indexes, counts) of the results
            cap = self.shape[0]
            idx, cnt = np.zeros(1 + cap, dtype=np.uint32), np.zeros(max(cap, 1), dtype=np.uint64)
            L.check(self.ctx.lib.fbk_query_read(self.ctx.h, self.h, idx.ctypes.data, cnt.ctypes.data))
            r = int(idx[0])
            return idx[1 : 1 + r].copy(), cnt[:r].copy()
        sums, counts = np.zeros(self.shape[0], dtype=np.int64), np.zeros(self.shape[0], dtype=np.uint64)
        L.check(self.ctx.lib.fbk_query_read(self.ctx.h, self.h, sums.ctypes.data, counts.ctypes.data))
        return sums, counts

    def output(self) -> "Batch":
        """The output batch of the last run (row-valued queries): BORROWED — do not free, valid until the next run."""
        h = C.c_void_p()
        L.check(self.ctx.lib.fbk_query_output(self.ctx.h, self.h, C.byref(h)))
        b = Batch(self.ctx, h.value)
        b.owned = False  # free() on the wrapper is a no-op: the query owns the batch
        return b

    def free(self) -> None:
        if self.h:
            L.check(self.ctx.lib.fbk_query_free(self.ctx.h, self.h))
            self.h = C.c_void_p(None)


class Context:
    """One GPU.  Fails loudly when libfbk.so is missing or no gfx950 device is visible."""

    def __init__(self, device: int = 0, _handle: Optional[int] = None, _borrowed: bool = False):
        self.lib = L.load()
        self.borrowed = _borrowed  # a group member: closed by the group
        if _handle is not None:
            self.h = C.c_void_p(_handle)
            return
        h = C.c_void_p()
        L.check(self.lib.fbk_open(device, 0, C.byref(h)))
        self.h = h

    def close(self) -> None:
        if self.h and not self.borrowed:
            L.check(self.lib.fbk_close(self.h))
        self.h = C.c_void_p(None)

    def fork(self) -> "Context":
        """A second context on the same device (own stream / lock / pool) for another calling
        thread; shares the fragment cache with this one (fbk_ctx_fork)."""
        h = C.c_void_p()
        L.check(self.lib.fbk_ctx_fork(self.h, C.byref(h)))
        return Context(_handle=h.value)

    def set_option(self, name: str, value: int) -> None:
        L.check(self.lib.fbk_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        L.check(self.lib.fbk_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def last_error(self) -> Tuple[int, str]:
        """(status code, message) of the last failing call on THIS context (fbk_last_error_r)."""
        buf = C.create_string_buffer(512)
        code = C.c_int32()
        L.check(self.lib.fbk_last_error_r(self.h, buf, 512, C.byref(code)))
        return int(code.value), buf.value.decode()

    def set_stream(self, hip_stream: int) -> None:
        L.check(self.lib.fbk_set_stream(self.h, C.c_void_p(hip_stream or None)))

    # ---- one process per GPU: the exchange issued by the library (fbk_comm_*) -------------------
    def comm_unique_id(self) -> bytes:
        """rank 0: the 128 bytes every rank passes to comm_init (hand them over with whatever channel the host has)"""
        buf = (C.c_uint8 * L.COMM_ID_BYTES)()
        L.check(self.lib.fbk_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid: bytes, n_ranks: int, rank: int) -> None:
        """collective: returns when every rank has called it"""
        assert len(uid) == L.COMM_ID_BYTES
        buf = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(uid)
        L.check(self.lib.fbk_comm_init(self.h, buf, n_ranks, rank))

    def comm_all_reduce(self, device_ptr: int, n_words: int) -> None:
        """asynchronous in-place sum of n_words uint64 over the ranks: after what the context's stream holds, on the
        communicator's stream; the words must not be touched before comm_fence()"""
        L.check(self.lib.fbk_comm_all_reduce_u64(self.h, C.c_void_p(device_ptr), n_words))

    def comm_fence(self) -> None:
        L.check(self.lib.fbk_comm_fence(self.h))

    def comm_close(self) -> None:
        L.check(self.lib.fbk_comm_close(self.h))

    def synchronize(self) -> None:
        L.check(self.lib.fbk_synchronize(self.h))

    # -- residency ------------------------------------------------------------------
    def upload(self, rows: Sequence[Row]) -> Batch:
        """rows[i] maps container key -> Container (what fragment.row() yields)."""
        n_desc = sum(len(r) for r in rows)
        descs = (L.ContainerDesc * max(n_desc, 1))()
        chunks: List[bytes] = []
        off = 0
        i = 0
        for ri, row in enumerate(rows):
            for key, c in row.items():
                d = descs[i]
                d.key, d.off, d.row, d.len, d.n, d.type = key, off, ri, c.length, c.n, c.typ
                b = c.payload()
                chunks.append(b)
                off += len(b)
                i += 1
        payload = b"".join(chunks) or b"\0"
        h = C.c_void_p()
        L.check(self.lib.fbk_batch_upload(self.h, descs, n_desc, len(rows), payload, off, C.byref(h)))
        return Batch(self, h.value)

    def upload_flat(self, descs: np.ndarray, payload: np.ndarray, n_rows: int) -> Batch:
        """The flattened form directly: `descs` a structured array laid out as fbk_container_desc
        (32 bytes per record), `payload` the bytes its offsets point into."""
        d = np.ascontiguousarray(descs)
        assert d.dtype.itemsize == C.sizeof(L.ContainerDesc)
        pay = np.ascontiguousarray(payload).view(np.uint8).reshape(-1)
        h = C.c_void_p()
        L.check(
            self.lib.fbk_batch_upload(
                self.h, C.cast(d.ctypes.data, C.POINTER(L.ContainerDesc)), d.size, n_rows, pay.ctypes.data, pay.size, C.byref(h)
            )
        )
        return Batch(self, h.value)

    def upload_roaring(self, data: bytes) -> Tuple[Batch, np.ndarray]:
        """A serialised roaring bitmap (Pilosa or official format) -> (batch, row ids): batch row
        i holds the containers with key >> 4 == row_ids[i] (Bitmap.UnmarshalBinary, roaring.go:1945)."""
        n = C.c_uint32()
        h = C.c_void_p()
        cap = max(len(data) // 2, 1)  # every container takes at least 2 payload bytes
        ids = np.zeros(cap, dtype=np.uint64)
        L.check(self.lib.fbk_batch_upload_roaring(self.h, data, len(data), C.byref(h), ids.ctypes.data, cap, C.byref(n)))
        return Batch(self, h.value), ids[: n.value].copy()

    def rbf_find_root(self, file_bytes: bytes, name: str) -> int:
        pg = C.c_uint32()
        L.check(self.lib.fbk_rbf_find_root(file_bytes, len(file_bytes), name.encode(), C.byref(pg)))
        return pg.value

    def upload_rbf(self, file_bytes: bytes, root_pgno: int) -> Tuple[Batch, np.ndarray]:
        """One bitmap b-tree of an RBF file image -> (batch, row ids) (rbf/tx.go ContainerIterator)."""
        n = C.c_uint32()
        h = C.c_void_p()
        cap = max(len(file_bytes) // 20, 1)  # a leaf cell takes at least 18 + 2 bytes
        ids = np.zeros(cap, dtype=np.uint64)
        L.check(self.lib.fbk_batch_upload_rbf(self.h, file_bytes, len(file_bytes), root_pgno, C.byref(h), ids.ctypes.data, cap, C.byref(n)))
        return Batch(self, h.value), ids[: n.value].copy()

    def upload_dense_device(self, device_ptr: int, n_rows: int) -> Batch:
        """n_rows dense rows (16 x 1024 uint64 each) that already sit in this device's memory at device_ptr
        (fbk_batch_upload_dense accepts a device source: the copy is device to device)."""
        h = C.c_void_p()
        L.check(self.lib.fbk_batch_upload_dense(self.h, C.c_void_p(device_ptr), n_rows, C.byref(h)))
        return Batch(self, h.value)

    def upload_dense(self, words: np.ndarray) -> Batch:
        w = np.ascontiguousarray(words, dtype=np.uint64)
        assert w.size % (SLOTS * BITMAP_WORDS) == 0
        n_rows = w.size // (SLOTS * BITMAP_WORDS)
        h = C.c_void_p()
        L.check(self.lib.fbk_batch_upload_dense(self.h, w.ctypes.data, n_rows, C.byref(h)))
        return Batch(self, h.value)

    # -- one-shot ops -----------------------------------------------------------------
    # -- device fragment cache ------------------------------------------------------------
    def cache_put(self, key: str, version: int, batch: Batch, row_ids) -> None:
        """The cache takes ownership of `batch` (do not free it)."""
        ids = np.ascontiguousarray(row_ids, dtype=np.uint64)
        L.check(self.lib.fbk_cache_put(self.h, key.encode(), version, batch.h, ids.ctypes.data, ids.size))
        batch.owned = False

    def cache_get(self, key: str, version: int) -> Optional[Tuple[Batch, np.ndarray]]:
        """Pinned (batch, row ids) or None on a miss; call cache_release(batch) when done."""
        h, ids, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
        rc = self.lib.fbk_cache_get(self.h, key.encode(), version, C.byref(h), C.byref(ids), C.byref(n))
        if rc == L.FBK_E_NOTFOUND:
            return None
        L.check(rc)
        b = Batch(self, h.value)
        b.owned = False
        rid = np.ctypeslib.as_array(C.cast(ids, C.POINTER(C.c_uint64)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)
        return b, rid

    def cache_release(self, batch: Batch) -> None:
        L.check(self.lib.fbk_cache_release(self.h, batch.h))

    def cache_invalidate(self, key_prefix: str) -> int:
        n = C.c_uint32()
        L.check(self.lib.fbk_cache_invalidate(self.h, key_prefix.encode(), C.byref(n)))
        return n.value

    def cache_configure(self, cap_bytes: int) -> None:
        L.check(self.lib.fbk_cache_configure(self.h, cap_bytes))

    def cache_stats(self) -> dict:
        v = [C.c_uint64() for _ in range(5)]
        L.check(self.lib.fbk_cache_stats(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("entries", "bytes", "hits", "misses", "evictions"), (x.value for x in v)))

    def intersection_count(self, a: Batch, rows_a, b: Batch, rows_b) -> np.ndarray:
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        rb = np.ascontiguousarray(rows_b, dtype=np.uint32)
        assert ra.size == rb.size
        out = np.zeros(ra.size, dtype=np.uint64)
        L.check(self.lib.fbk_intersection_count(self.h, a.h, ra.ctypes.data, b.h, rb.ctypes.data, ra.size, out.ctypes.data))
        return out

    def setop(self, op: int, a: Batch, rows_a, b: Batch, rows_b, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        rb = np.ascontiguousarray(rows_b, dtype=np.uint32)
        assert ra.size == rb.size
        out = np.zeros(ra.size, dtype=np.uint64)
        h = C.c_void_p()
        L.check(
            self.lib.fbk_setop(self.h, op, a.h, ra.ctypes.data, b.h, rb.ctypes.data, ra.size, flags, C.byref(h), out.ctypes.data)
        )
        return Batch(self, h.value), out

    # -- n-way union ------------------------------------------------------------------
    def union_n(self, batch: Batch, groups, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        """groups: array [n_groups, k] of row ordinals; out row g = union of group g."""
        g = np.ascontiguousarray(groups, dtype=np.uint32).reshape(len(groups), -1) if len(groups) else np.zeros((0, 0), np.uint32)
        n_groups, k = g.shape
        out = np.zeros(n_groups, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self.lib.fbk_union_n(self.h, batch.h, g.ctypes.data, n_groups, k, flags, C.byref(h), out.ctypes.data))
        return Batch(self, h.value), out

    def union_n_intersection_count(self, batch: Batch, groups, filt: Optional[Batch] = None, rows_f=None) -> np.ndarray:
        g = np.ascontiguousarray(groups, dtype=np.uint32).reshape(len(groups), -1)
        n_groups, k = g.shape
        out = np.zeros(n_groups, dtype=np.uint64)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        L.check(
            self.lib.fbk_union_n_intersection_count(
                self.h, batch.h, g.ctypes.data, n_groups, k, filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None, out.ctypes.data
            )
        )
        return out

    def fold_n(self, op: int, batch: Batch, groups, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        """groups: [n_groups, k] row ordinals; out row g = r0 <op> r1 <op> ... folded left to right
        (executeIntersect/Union/Xor/DifferenceShard, executor.go:5357, 5382, 5513, 2950)."""
        g = np.ascontiguousarray(groups, dtype=np.uint32).reshape(len(groups), -1) if len(groups) else np.zeros((0, 0), np.uint32)
        n_groups, k = g.shape
        out = np.zeros(n_groups, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self.lib.fbk_fold_n(self.h, op, batch.h, g.ctypes.data, n_groups, k, flags, C.byref(h), out.ctypes.data))
        return Batch(self, h.value), out

    def fold_n_intersection_count(self, op: int, batch: Batch, groups, filt: Optional[Batch] = None, rows_f=None) -> np.ndarray:
        g = np.ascontiguousarray(groups, dtype=np.uint32).reshape(len(groups), -1)
        n_groups, k = g.shape
        out = np.zeros(n_groups, dtype=np.uint64)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        L.check(
            self.lib.fbk_fold_n_intersection_count(
                self.h, op, batch.h, g.ctypes.data, n_groups, k, filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None, out.ctypes.data
            )
        )
        return out

    def topk(self, a: Batch, rows_a, k: int = 0, filt: Optional[Batch] = None, rows_f=None) -> Tuple[np.ndarray, np.ndarray]:
        """rows_a: [n_shards, n_a]; (row indices, counts) of the k rows with the largest
        |row ∩ filter| summed over the shards (count descending, index ascending; zeros dropped)."""
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        n_shards, n_a = ra.shape
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        cap = n_a if k == 0 else min(k, n_a)
        idx = np.zeros(max(cap, 1), dtype=np.uint32)
        cnt = np.zeros(max(cap, 1), dtype=np.uint64)
        n = C.c_uint32()
        L.check(
            self.lib.fbk_topk(
                self.h, a.h, ra.ctypes.data, n_a, filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None,
                n_shards, k, idx.ctypes.data, cnt.ctypes.data, cap, C.byref(n),
            )
        )
        return idx[: n.value].copy(), cnt[: n.value].copy()

    def topn(self, a: Batch, rows_a, n: int = 0, filt: Optional[Batch] = None, rows_f=None, min_threshold: int = 0,
             tanimoto_threshold: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """fbk_topn: topk with fragment.top's MinThreshold / TanimotoThreshold rules (fragment.go:1317)."""
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        n_shards, n_a = ra.shape
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        cap = n_a if n == 0 else min(n, n_a)
        idx = np.zeros(max(cap, 1), dtype=np.uint32)
        cnt = np.zeros(max(cap, 1), dtype=np.uint64)
        got = C.c_uint32()
        L.check(
            self.lib.fbk_topn(
                self.h, a.h, ra.ctypes.data, n_a, filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None,
                n_shards, n, min_threshold, tanimoto_threshold, idx.ctypes.data, cnt.ctypes.data, cap, C.byref(got),
            )
        )
        return idx[: got.value].copy(), cnt[: got.value].copy()

    def topn_partials(self, a: Optional[Batch], rows_a, n_a: int, n: int = 0, filt: Optional[Batch] = None, rows_f=None, min_threshold: int = 0,
                      tanimoto_threshold: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """fbk_topn_partials: (totals [n_a], candidate flags [n_a]) of THIS context's shards — what a rank of the
        one-process-per-GPU deployment hands to featurebase_amd.dist.topn_reduce.  a = None / no rows: zeros."""
        tot, cand = np.zeros(n_a, dtype=np.uint64), np.zeros(n_a, dtype=np.uint64)
        if a is None or rows_a is None or len(rows_a) == 0:
            return tot, cand
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        assert ra.ndim == 2 and ra.shape[1] == n_a
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        L.check(self.lib.fbk_topn_partials(self.h, a.h, ra.ctypes.data, n_a, filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None,
                                           ra.shape[0], n, min_threshold, tanimoto_threshold, tot.ctypes.data, cand.ctypes.data))
        return tot, cand

    def topk_bsi(self, a: Batch, rows_a, filt: Optional[Batch] = None, rows_f=None, flags: int = 0) -> Tuple[Batch, int]:
        """The TopK counts as BSI planes over the row indices (bsiBuilder, bsi.go:251): (batch of depth rows, depth)."""
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        n_shards, n_a = ra.shape
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        h, depth = C.c_void_p(), C.c_uint32()
        L.check(self.lib.fbk_topk_bsi(self.h, a.h, ra.ctypes.data, n_a, filt.h if filt is not None else None,
                                      rf.ctypes.data if rf is not None else None, n_shards, flags, C.byref(h), C.byref(depth)))
        return Batch(self, h.value), int(depth.value)

    def flip(self, batch: Batch, rows, start: int, end: int, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        """out row i = rows[i] with the bits of [start, end] (inclusive, row-relative) negated: Bitmap.Flip (roaring.go:2769)."""
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros(r.size, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self.lib.fbk_flip(self.h, batch.h, r.ctypes.data, r.size, start, end, flags, C.byref(h), out.ctypes.data))
        return Batch(self, h.value), out

    NO_COLUMN = 0xFFFFFFFFFFFFFFFF

    def rows(self, batch: Batch, rows, column: Optional[int] = None, limit: int = 0, row_ids=None) -> np.ndarray:
        """fragment.rows (fragment.go:2465): positions (into `rows`, the fragment's rows in ascending row-id
        order) of the rows that hold anything / hold `column`, under the reference's limit rule (row_ids: the
        fragment row ids of `rows`, needed when a column and a limit are both given)."""
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        ids = np.ascontiguousarray(row_ids, dtype=np.uint64) if row_ids is not None else None
        if ids is not None and ids.size != r.size:
            raise ValueError("row_ids must name every row")
        out = np.zeros(max(r.size, 1), dtype=np.uint32)
        n = C.c_uint64()
        L.check(self.lib.fbk_rows(self.h, batch.h, r.ctypes.data, ids.ctypes.data if ids is not None else None, r.size,
                                  self.NO_COLUMN if column is None else int(column), int(limit), out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value].copy()

    NO_ROW = 0xFFFFFFFF

    def shift(self, batch: Batch, rows, carry_rows=None, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        """out row i = rows[i] shifted up by one column, column 0 taken from the last column of
        carry_rows[i] (the previous shard's row); NO_ROW = absent (Row.Shift, row.go:374)."""
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        cr = np.ascontiguousarray(carry_rows, dtype=np.uint32) if carry_rows is not None else None
        assert cr is None or cr.size == r.size
        out = np.zeros(r.size, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self.lib.fbk_shift(self.h, batch.h, r.ctypes.data, cr.ctypes.data if cr is not None else None, r.size, flags, C.byref(h), out.ctypes.data))
        return Batch(self, h.value), out

    def count_range(self, batch: Batch, rows, start: int, end: int) -> np.ndarray:
        """out[i] = bits of rows[i] in [start, end), positions relative to the row (0..2^20):
        Bitmap.CountRange (roaring.go:573)."""
        r = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros(r.size, dtype=np.uint64)
        L.check(self.lib.fbk_count_range(self.h, batch.h, r.ctypes.data, r.size, start, end, out.ctypes.data))
        return out

    # -- GroupBy / TopK count matrix -----------------------------------------------------
    def count_matrix(self, a: Batch, rows_a, b: Batch, rows_b, filt: Optional[Batch] = None, rows_f=None, per_shard: bool = False):
        """rows_a: [n_shards, n_a], rows_b: [n_shards, n_b], rows_f: [n_shards].
        Returns total [n_a, n_b] (and the per-shard [n_shards, n_a, n_b] if asked)."""
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        rb = np.ascontiguousarray(rows_b, dtype=np.uint32)
        n_shards, n_a = ra.shape
        n_b = rb.shape[1]
        assert rb.shape[0] == n_shards
        tot = np.zeros((n_a, n_b), dtype=np.uint64)
        ps = np.zeros((n_shards, n_a, n_b), dtype=np.uint64) if per_shard else None
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        L.check(
            self.lib.fbk_count_matrix(
                self.h, a.h, ra.ctypes.data, n_a, b.h, rb.ctypes.data, n_b,
                filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None,
                n_shards, tot.ctypes.data, ps.ctypes.data if ps is not None else None,
            )
        )
        return (tot, ps) if per_shard else tot

    # -- prepared (launch-only) forms of the query-level calls -------------------------------
    def prepare_count_matrix(self, a: Batch, rows_a, b: Batch, rows_b, filt: Optional[Batch] = None, rows_f=None, keep_per_shard: bool = False) -> Query:
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        rb = np.ascontiguousarray(rows_b, dtype=np.uint32)
        n_shards, n_a = ra.shape
        n_b = rb.shape[1]
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        h = C.c_void_p()
        L.check(self.lib.fbk_query_count_matrix(self.h, a.h, ra.ctypes.data, n_a, b.h, rb.ctypes.data, n_b, filt.h if filt is not None else None,
                                                rf.ctypes.data if rf is not None else None, n_shards, 1 if keep_per_shard else 0, C.byref(h)))
        return Query(self, h.value, "count_matrix", (n_shards, n_a, n_b))

    def prepare_fold_intersection_count(self, op: int, batch: Batch, groups, filt: Optional[Batch] = None, rows_f=None) -> Query:
        g = np.ascontiguousarray(groups, dtype=np.uint32)
        g = g.reshape(len(groups), -1)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        h = C.c_void_p()
        L.check(self.lib.fbk_query_fold_intersection_count(self.h, op, batch.h, g.ctypes.data, g.shape[0], g.shape[1], filt.h if filt is not None else None,
                                                           rf.ctypes.data if rf is not None else None, C.byref(h)))
        return Query(self, h.value, "fold", (g.shape[0],))

    def prepare_bsi_sum(self, batch: Batch, base_rows, bit_depth: int, op: int = 0, predicate: int = 0, filt: Optional[Batch] = None, rows_f=None) -> Query:
        """op == 0: Sum over exists ∩ filt; op = BSI_*: Sum(Row(v op predicate), field = v) in one pass"""
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        h = C.c_void_p()
        L.check(self.lib.fbk_query_bsi_sum(self.h, batch.h, base.ctypes.data, base.size, bit_depth, op, C.c_int64(predicate), filt.h if filt is not None else None,
                                           rf.ctypes.data if rf is not None else None, C.byref(h)))
        return Query(self, h.value, "bsi", (base.size,))

    def prepare_bsi_range(self, batch: Batch, base_rows, op: int, bit_depth: int, predicate: int) -> Query:
        """Row(v op predicate) with the result rows kept on the device (Query.output()); read() = their cardinalities"""
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        h = C.c_void_p()
        L.check(self.lib.fbk_query_bsi_range(self.h, batch.h, base.ctypes.data, base.size, op, bit_depth, C.c_int64(predicate), C.byref(h)))
        return Query(self, h.value, "rows", (base.size,))

    def prepare_fold(self, op: int, batch: Batch, groups, flags: int = 0) -> Query:
        """The materialised n-way Union / Xor / Difference (fbk_fold_n) as a launch-only query; read() = cardinalities"""
        g = np.ascontiguousarray(groups, dtype=np.uint32)
        h = C.c_void_p()
        L.check(self.lib.fbk_query_fold(self.h, op, batch.h, g.ctypes.data, g.shape[0], g.shape[1], flags, C.byref(h)))
        return Query(self, h.value, "rows", (g.shape[0],))

    def prepare_topn(self, a: Batch, rows_a, n: int = 0, filt: Optional[Batch] = None, rows_f=None, min_threshold: int = 0, tanimoto_threshold: int = 0) -> Query:
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        n_shards, n_a = ra.shape
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        h = C.c_void_p()
        L.check(self.lib.fbk_query_topn(self.h, a.h, ra.ctypes.data, n_a, filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None, n_shards, n,
                                        min_threshold, tanimoto_threshold, C.byref(h)))
        return Query(self, h.value, "topn", (min(n, n_a) if n else n_a,))

    # -- BSI -----------------------------------------------------------------------------
    def bsi_sum(self, batch: Batch, base_rows, bit_depth: int, filt: Optional[Batch] = None, rows_f=None):
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        sums = np.zeros(base.size, dtype=np.int64)
        counts = np.zeros(base.size, dtype=np.uint64)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        L.check(
            self.lib.fbk_bsi_sum(
                self.h, batch.h, base.ctypes.data, base.size, bit_depth,
                filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None, sums.ctypes.data, counts.ctypes.data,
            )
        )
        return sums, counts

    def _bsi_minmax(self, fn, batch: Batch, base_rows, bit_depth: int, filt: Optional[Batch], rows_f):
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        vals = np.zeros(base.size, dtype=np.int64)
        counts = np.zeros(base.size, dtype=np.uint64)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        L.check(
            fn(
                self.h, batch.h, base.ctypes.data, base.size, bit_depth,
                filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None, vals.ctypes.data, counts.ctypes.data,
            )
        )
        return vals, counts

    def bsi_min(self, batch: Batch, base_rows, bit_depth: int, filt: Optional[Batch] = None, rows_f=None):
        """Per shard (min, count): fragment.min (fragment.go:754)."""
        return self._bsi_minmax(self.lib.fbk_bsi_min, batch, base_rows, bit_depth, filt, rows_f)

    def bsi_max(self, batch: Batch, base_rows, bit_depth: int, filt: Optional[Batch] = None, rows_f=None):
        """Per shard (max, count): fragment.max (fragment.go:803)."""
        return self._bsi_minmax(self.lib.fbk_bsi_max, batch, base_rows, bit_depth, filt, rows_f)

    def bsi_distinct(self, batch: Batch, base_rows, bit_depth: int, filt: Optional[Batch] = None, rows_f=None) -> np.ndarray:
        """Sorted distinct stored values (int64, Base not added) of the columns in exists ∩ filter
        over all given shards (executeDistinctShardBSI, executor.go:2034)."""
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        n = C.c_uint64()
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=np.int64)
            rc = self.lib.fbk_bsi_distinct(self.h, batch.h, base.ctypes.data, base.size, bit_depth, filt.h if filt is not None else None,
                                           rf.ctypes.data if rf is not None else None, out.ctypes.data, cap, C.byref(n))
            if rc == L.FBK_E_CAPACITY:
                cap = int(n.value)
                continue
            L.check(rc)
            return out[: n.value].copy()

    def bsi_add(self, x: Batch, rows_x, y: Batch, rows_y, flags: int = 0) -> Batch:
        """rows_x: [n_groups, depth_x], rows_y: [n_groups, depth_y] plane rows (bit i = column i);
        out rows g*(D+1)+i = plane i of x + y (roaring.Add, roaring/add.go:12)."""
        rx = np.ascontiguousarray(rows_x, dtype=np.uint32).reshape(len(rows_x), -1)
        ry = np.ascontiguousarray(rows_y, dtype=np.uint32).reshape(len(rows_y), -1)
        assert rx.shape[0] == ry.shape[0]
        h = C.c_void_p()
        L.check(self.lib.fbk_bsi_add(self.h, x.h, rx.ctypes.data, rx.shape[1], y.h, ry.ctypes.data, ry.shape[1], rx.shape[0], flags, C.byref(h)))
        return Batch(self, h.value)

    def bsi_range(self, batch: Batch, base_rows, op: int, bit_depth: int, predicate: int, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        counts = np.zeros(base.size, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self.lib.fbk_bsi_range(self.h, batch.h, base.ctypes.data, base.size, op, bit_depth, predicate, flags, C.byref(h), counts.ctypes.data))
        return Batch(self, h.value), counts

    def bsi_range_sum(self, batch: Batch, base_rows, op: int, bit_depth: int, predicate: int, filt: Optional[Batch] = None, rows_f=None):
        """Sum(Row(v op predicate), field = v) in one pass over the planes: (sums int64[n_shards], counts uint64[n_shards]),
        equal to bsi_sum with filter = bsi_range(op, predicate) (∩ filt)."""
        b = np.ascontiguousarray(base_rows, dtype=np.uint32)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        sums = np.zeros(b.size, dtype=np.int64)
        counts = np.zeros(b.size, dtype=np.uint64)
        L.check(self.lib.fbk_bsi_range_sum(self.h, batch.h, b.ctypes.data, b.size, op, bit_depth, C.c_int64(predicate),
                                           filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None,
                                           sums.ctypes.data, counts.ctypes.data))
        return sums, counts

    def bsi_range_between_sum(self, batch: Batch, base_rows, bit_depth: int, lo: int, hi: int, filt: Optional[Batch] = None, rows_f=None):
        """Sum(Row(lo <= v <= hi), field = v): (sums, counts) per shard, equal to bsi_sum over bsi_range_between (∩ filt)."""
        b = np.ascontiguousarray(base_rows, dtype=np.uint32)
        rf = np.ascontiguousarray(rows_f, dtype=np.uint32) if filt is not None else None
        sums = np.zeros(b.size, dtype=np.int64)
        counts = np.zeros(b.size, dtype=np.uint64)
        L.check(self.lib.fbk_bsi_range_between_sum(self.h, batch.h, b.ctypes.data, b.size, bit_depth, C.c_int64(lo), C.c_int64(hi),
                                                   filt.h if filt is not None else None, rf.ctypes.data if rf is not None else None,
                                                   sums.ctypes.data, counts.ctypes.data))
        return sums, counts

    def bsi_range_between(self, batch: Batch, base_rows, bit_depth: int, lo: int, hi: int, flags: int = 0) -> Tuple[Batch, np.ndarray]:
        base = np.ascontiguousarray(base_rows, dtype=np.uint32)
        counts = np.zeros(base.size, dtype=np.uint64)
        h = C.c_void_p()
        L.check(self.lib.fbk_bsi_range_between(self.h, batch.h, base.ctypes.data, base.size, bit_depth, lo, hi, flags, C.byref(h), counts.ctypes.data))
        return Batch(self, h.value), counts

    def plan(self, a: Batch, rows_a, b: Batch, rows_b, device_counts_ptr: int = 0) -> Plan:
        ra = np.ascontiguousarray(rows_a, dtype=np.uint32)
        rb = np.ascontiguousarray(rows_b, dtype=np.uint32)
        assert ra.size == rb.size
        h = C.c_void_p()
        L.check(
            self.lib.fbk_plan_create(
                self.h, a.h, ra.ctypes.data, b.h, rb.ctypes.data, ra.size, C.c_void_p(device_counts_ptr or None), C.byref(h)
            )
        )
        return Plan(self, h.value, int(ra.size))


class Group:
    """Several GPUs behind one process (fbk_group_*): member m owns the shards s with s % G == m;
    partial counts are reduced inside the library (mapReduce + reduceFn, executor.go:6449-6533)."""

    def __init__(self, devices: Sequence[int]):
        self.lib = L.load()
        dv = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        L.check(self.lib.fbk_group_open(dv, len(devices), 0, C.byref(h)))
        self.h = h
        self.members: List[Context] = []
        for i in range(len(devices)):
            m = C.c_void_p()
            L.check(self.lib.fbk_group_member(self.h, i, C.byref(m)))
            self.members.append(Context(_handle=m.value, _borrowed=True))

    def __len__(self) -> int:
        return len(self.members)

    def set_reduce(self, mode: int) -> None:
        L.check(self.lib.fbk_group_set_reduce(self.h, mode))

    def plan_intersection_count_total(self, plans: Sequence[Optional[Plan]]) -> int:
        arr = (C.c_void_p * len(self.members))(*[(p.h if p is not None else None) for p in plans])
        tot = C.c_uint64()
        L.check(self.lib.fbk_group_plan_intersection_count_total(self.h, arr, C.byref(tot)))
        return int(tot.value)

    def count_matrix(self, per_member: Sequence[Optional[dict]], n_a: int, n_b: int) -> np.ndarray:
        """per_member[m]: None or dict(a=Batch, rows_a=[n_shards, n_a], b=Batch, rows_b=[n_shards, n_b],
        filt=Batch|None, rows_f=[n_shards])."""
        args = (L.MatrixArgs * len(self.members))()
        keep = []
        for m, pm in enumerate(per_member):
            if pm is None:
                continue
            ra = np.ascontiguousarray(pm["rows_a"], dtype=np.uint32)
            rb = np.ascontiguousarray(pm["rows_b"], dtype=np.uint32)
            assert ra.shape[1] == n_a and rb.shape[1] == n_b and ra.shape[0] == rb.shape[0]
            rf = np.ascontiguousarray(pm["rows_f"], dtype=np.uint32) if pm.get("filt") is not None else None
            keep += [ra, rb, rf]
            args[m].a, args[m].rows_a = pm["a"].h, ra.ctypes.data
            args[m].b, args[m].rows_b = pm["b"].h, rb.ctypes.data
            args[m].filter = pm["filt"].h if pm.get("filt") is not None else None
            args[m].rows_f = rf.ctypes.data if rf is not None else None
            args[m].n_shards = ra.shape[0]
        tot = np.zeros((n_a, n_b), dtype=np.uint64)
        L.check(self.lib.fbk_group_count_matrix(self.h, args, n_a, n_b, tot.ctypes.data))
        return tot

    def bsi_sum(self, per_member: Sequence[Optional[dict]], bit_depth: int) -> Tuple[int, int]:
        """per_member[m]: None or dict(batch=Batch, base_rows=[n_shards], filt=Batch|None, rows_f=[n_shards]).
        Returns (sum, count) over every shard of every member (fbk_group_bsi_sum)."""
        args = (L.BsiArgs * len(self.members))()
        keep = []
        for m, pm in enumerate(per_member):
            if pm is None:
                continue
            base = np.ascontiguousarray(pm["base_rows"], dtype=np.uint32)
            rf = np.ascontiguousarray(pm["rows_f"], dtype=np.uint32) if pm.get("filt") is not None else None
            keep += [base, rf]
            args[m].batch, args[m].base_rows = pm["batch"].h, base.ctypes.data
            args[m].filter = pm["filt"].h if pm.get("filt") is not None else None
            args[m].rows_f = rf.ctypes.data if rf is not None else None
            args[m].n_shards = base.size
        s, c = C.c_int64(), C.c_uint64()
        L.check(self.lib.fbk_group_bsi_sum(self.h, args, bit_depth, C.byref(s), C.byref(c)))
        return int(s.value), int(c.value)

    def topn(self, per_member: Sequence[Optional[dict]], n_a: int, n: int = 0, min_threshold: int = 0, tanimoto_threshold: int = 0):
        """per_member[m]: None or dict(a=Batch, rows_a=[n_shards, n_a], filt=Batch|None, rows_f=[n_shards]).
        Returns (row indexes, counts) of the TopN over the shards of all members (fbk_group_topn: fbk_topn's answer,
        however the shards are dealt)."""
        args = (L.TopnArgs * len(self.members))()
        keep = []
        for m, pm in enumerate(per_member):
            if pm is None:
                continue
            ra = np.ascontiguousarray(pm["rows_a"], dtype=np.uint32)
            assert ra.ndim == 2 and ra.shape[1] == n_a
            rf = np.ascontiguousarray(pm["rows_f"], dtype=np.uint32) if pm.get("filt") is not None else None
            keep += [ra, rf]
            args[m].a, args[m].rows_a = pm["a"].h, ra.ctypes.data
            args[m].filter = pm["filt"].h if pm.get("filt") is not None else None
            args[m].rows_f = rf.ctypes.data if rf is not None else None
            args[m].n_shards = ra.shape[0]
        cap = n_a
        idx, cnt, got = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint64), C.c_uint32()
        L.check(self.lib.fbk_group_topn(self.h, args, n_a, n, min_threshold, tanimoto_threshold, idx.ctypes.data, cnt.ctypes.data, cap, C.byref(got)))
        return idx[: got.value].copy(), cnt[: got.value].copy()

    def reduce_u64(self, device_ptrs: Sequence[int], words: int) -> np.ndarray:
        arr = (C.c_void_p * len(self.members))(*[(p or None) for p in device_ptrs])
        out = np.zeros(words, dtype=np.uint64)
        L.check(self.lib.fbk_group_reduce_u64(self.h, arr, words, out.ctypes.data))
        return out

    def close(self) -> None:
        if self.h:
            L.check(self.lib.fbk_group_close(self.h))
            self.h = C.c_void_p(None)
            for m in self.members:
                m.h = C.c_void_p(None)
