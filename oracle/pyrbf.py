"""RBF page reader / writer — TEST INFRASTRUCTURE ONLY (see oracle/pyoracle.py).

Restates the on-disk layout of the reference's storage engine (rbf/rbf.go):
  page size 8192 (:29); meta page: magic "\\xFFRBF" (:26), pageN u32 @8, walID i64 @12, root
  record pgno u32 @20, freelist pgno u32 @24, all big endian (:120-141);
  root record page: header 12 bytes (overflow pgno u32 BE @8), records {pgno u32 BE, len u16 BE,
  name} (:156-255);
  branch / leaf page header: pgno u32, flags u32, cellN u16 big endian, then cellN u16 BE cell
  offsets (:189-216); cells start at align8(10 + 2n) and are 8-byte aligned;
  leaf cell: key u64, type u32, ElemN u16, BitN u32 little endian, data @18 (:488-520, 585-593):
  array = ElemN x u16, RLE = ElemN x {start u16, last u16}, BitmapPtr = pgno u32 of a raw 8 KiB
  bitmap page (:63-69, rbf/tx.go:1315-1321);
  branch cell: leftKey u64, flags u32, childPgno u32 little endian (:616-642).
Limits: arrays <= 4079 values, RLE <= 2039 runs per cell (:37-42); denser containers are
stored as bitmap pages (rbf/cursor.go:421-471).
Pinned by tests/test_oracle_rbf.py to the one RBF database the reference ships
(rbf/testdata/check/bad-freelist: a leaf page written by the reference itself)."""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

PAGE = 8192
MAGIC = b"\xffRBF"
LEAF, BRANCH, ROOT_RECORD = 2, 4, 1
T_ARRAY, T_RLE, T_BITMAP, T_BITMAP_PTR = 1, 2, 3, 4
ARRAY_MAX, RLE_MAX = 4079, 2039


def align8(x: int) -> int:
    return (x + 7) & ~7


# ---- reader ------------------------------------------------------------------------------------
def find_root(data: bytes, name: str) -> int:
    assert data[:4] == MAGIC
    pg = struct.unpack_from(">I", data, 20)[0]
    while pg:
        page = data[pg * PAGE : (pg + 1) * PAGE]
        pos = 12
        while pos + 6 <= PAGE:
            root = struct.unpack_from(">I", page, pos)[0]
            if root == 0:
                break
            sz = struct.unpack_from(">H", page, pos + 4)[0]
            pos += 6
            if page[pos : pos + sz].decode() == name:
                return root
            pos += sz
        pg = struct.unpack_from(">I", page, 8)[0]
    raise KeyError(name)


def read_bitmap(data: bytes, root: int) -> List[Tuple[int, int, int, np.ndarray]]:
    """-> [(key, fbk type {1 array, 2 bitmap, 3 run}, n, payload)] in key order
    (Tx.ContainerIterator + toContainer, rbf/tx.go:1333, rbf/cursorx.go:230-266)."""
    out: List[Tuple[int, int, int, np.ndarray]] = []

    def walk(pgno: int):
        page = data[pgno * PAGE : (pgno + 1) * PAGE]
        _, flags, cell_n = struct.unpack_from(">IIH", page, 0)
        offs = [struct.unpack_from(">H", page, 10 + 2 * i)[0] for i in range(cell_n)]
        if flags == BRANCH:
            for o in offs:
                walk(struct.unpack_from("<I", page, o + 12)[0])
            return
        assert flags == LEAF, flags
        for o in offs:
            key, typ, elem_n, bit_n = struct.unpack_from("<QIHI", page, o)
            d = o + 18
            if typ == T_ARRAY:
                out.append((key, 1, bit_n, np.frombuffer(page, dtype="<u2", count=elem_n, offset=d).copy()))
            elif typ == T_RLE:
                out.append((key, 3, bit_n, np.frombuffer(page, dtype="<u2", count=2 * elem_n, offset=d).reshape(-1, 2).copy()))
            elif typ == T_BITMAP_PTR:
                bp = struct.unpack_from("<I", page, d)[0]
                out.append((key, 2, bit_n, np.frombuffer(data, dtype="<u8", count=1024, offset=bp * PAGE).copy()))
            else:
                raise ValueError(f"invalid container type: {typ}")

    walk(root)
    return out


# ---- writer ------------------------------------------------------------------------------------
def _leaf_cell(key: int, fbk_type: int, n: int, payload: np.ndarray, alloc_page) -> bytes:
    if fbk_type == 1:
        body = np.ascontiguousarray(payload, dtype="<u2").tobytes()
        return struct.pack("<QIHI", key, T_ARRAY, len(body) // 2, n) + body
    if fbk_type == 3:
        body = np.ascontiguousarray(payload, dtype="<u2").tobytes()
        return struct.pack("<QIHI", key, T_RLE, len(body) // 4, n) + body
    pg = alloc_page(np.ascontiguousarray(payload, dtype="<u8").tobytes())
    return struct.pack("<QIHI", key, T_BITMAP_PTR, 0, n) + struct.pack("<I", pg)


def _pack_pages(cells: List[bytes], flags: int, max_cells: Optional[int] = None) -> List[Tuple[List[bytes]]]:
    """greedy packing of cells into pages: cells at 8-byte aligned offsets after the index"""
    pages, cur = [], []
    for c in cells:
        trial = cur + [c]
        off = align8(10 + 2 * len(trial))
        for x in trial:
            off = align8(off) + len(x)
        if (off > PAGE or (max_cells and len(trial) > max_cells)) and cur:
            pages.append(cur)
            cur = [c]
        else:
            cur = trial
    if cur:
        pages.append(cur)
    return pages


def _render(pgno: int, flags: int, cells: List[bytes]) -> bytes:
    page = bytearray(PAGE)
    struct.pack_into(">IIH", page, 0, pgno, flags, len(cells))
    off = align8(10 + 2 * len(cells))
    for i, c in enumerate(cells):
        off = align8(off)
        struct.pack_into(">H", page, 10 + 2 * i, off)
        page[off : off + len(c)] = c
        off += len(c)
    assert off <= PAGE
    return bytes(page)


def write_db(bitmaps: Dict[str, List[Tuple[int, int, int, np.ndarray]]], leaf_cells_per_page: Optional[int] = None,
             branch_fanout: int = 64) -> bytes:
    """bitmaps: name -> [(key, fbk type, n, payload)] (keys ascending) -> an RBF file image:
    meta page 0, root record page 1, then each bitmap's bitmap pages, leaves and branches."""
    pages: List[Optional[bytes]] = [None, None]

    def alloc(raw: Optional[bytes] = None) -> int:
        pages.append(raw)
        return len(pages) - 1

    roots = {}
    for name, conts in bitmaps.items():
        cells = []
        for key, t, n, payload in conts:
            if t == 1:
                assert len(payload) <= ARRAY_MAX
            if t == 3:
                assert len(payload) <= RLE_MAX
            cells.append((key, _leaf_cell(key, t, n, payload, alloc)))
        groups = _pack_pages([c for _, c in cells], LEAF, leaf_cells_per_page) or [[]]
        level, i = [], 0
        for g in groups:
            pg = alloc()
            pages[pg] = _render(pg, LEAF, g)
            level.append((cells[i][0] if g else 0, pg))
            i += len(g)
        while len(level) > 1:  # branch levels
            nxt = []
            for j in range(0, len(level), branch_fanout):
                grp = level[j : j + branch_fanout]
                pg = alloc()
                pages[pg] = _render(pg, BRANCH, [struct.pack("<QII", k, 0, child) for k, child in grp])
                nxt.append((grp[0][0], pg))
            level = nxt
        roots[name] = level[0][1]
    meta = bytearray(PAGE)
    meta[:4] = MAGIC
    struct.pack_into(">I", meta, 8, len(pages))
    struct.pack_into(">q", meta, 12, 1)
    struct.pack_into(">I", meta, 20, 1)
    pages[0] = bytes(meta)
    rr = bytearray(PAGE)
    struct.pack_into(">II", rr, 0, 1, ROOT_RECORD)
    pos = 12
    for name in sorted(roots):
        nb = name.encode()
        struct.pack_into(">IH", rr, pos, roots[name], len(nb))
        rr[pos + 6 : pos + 6 + len(nb)] = nb
        pos += 6 + len(nb)
    pages[1] = bytes(rr)
    return b"".join(pages)


# ---- the write-side POLICY of the reference's cursor: which cell a container becomes -------------------------------
# (restated so that RLE cells, BitmapPtr cells and merged cells of this oracle's images are what the reference itself
# would have written for the same roaring containers; pinned by the expectations of rbf/cursor_test.go:290-440,
# 601-640 in tests/test_oracle_rbf.py)
def convert_to_leaf(key: int, c):
    """ConvertToLeafArgs (rbf/cursor.go:1299-1341): oracle container -> (key, fbk type, n, payload) or None when
    empty.  Arrays of more than ArrayMaxSize values and run containers of more than RLEMaxSize runs become bitmaps
    (:1309-1315, :1327-1334); bitmaps are stored as BitmapPtr cells by putLeafCell (:421-477)."""
    from . import pyoracle as O

    if c is None or not c.p or c.n == 0:
        return None
    if c.typ == O.ARRAY:
        if c.n > ARRAY_MAX:
            return (key, 2, c.n, c.words())
        return (key, 1, c.n, np.asarray(c.data(), dtype=np.uint16))
    if c.typ == O.RUN:
        if c.length > RLE_MAX:
            return (key, 2, c.n, c.words())
        return (key, 3, c.n, np.asarray(c.data(), dtype=np.uint16).reshape(-1, 2))
    return (key, 2, c.n, c.words())


def leaf_to_container(leaf):
    """toContainer / merge's reconstruction (rbf/cursorx.go:230-266, cursor.go:1352-1366)"""
    from . import pyoracle as O

    _, t, n, payload = leaf
    if t == 1:
        return O.OContainer.array(payload)
    if t == 3:
        return O.OContainer.run([tuple(x) for x in np.asarray(payload).reshape(-1, 2).tolist()])
    return O.OContainer.bitmap(np.asarray(payload, dtype=np.uint64), n)


class CursorModel:
    """The leaf cells of one RBF bitmap as Cursor.AddRoaring leaves them (rbf/cursor.go:1376-1408): a container
    under a new key is converted and inserted; under an existing key it is merged — roaring.Union (union +
    optimize(), roaring.go:7610-7615) of the incoming container and the stored one, rewritten only when the union has
    more bits than the incoming container (`res.N() != data.N()`, :1367 — the reference compares with the INCOMING
    container, so merging a subset of what is stored reports "changed" and rewrites the cell)."""

    def __init__(self):
        self.cells = {}

    def add_roaring(self, items) -> bool:
        """items: [(key, oracle container)] in key order -> changed"""
        from . import pyoracle as O

        changed = False
        for key, cont in items:
            leaf = convert_to_leaf(key, cont)
            if leaf is None:  # leaf.BitN == 0: skipped (:1381-1383)
                continue
            if key not in self.cells:
                self.cells[key] = leaf
                changed = True
                continue
            stored = leaf_to_container(self.cells[key])
            res = O.optimize(O.union(cont, stored))
            if res.n != cont.n:
                self.cells[key] = convert_to_leaf(key, res)
                changed = True
        return changed

    def containers(self):
        return [self.cells[k] for k in sorted(self.cells)]
