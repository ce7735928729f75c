"""RBF page WRITER — TEST INFRASTRUCTURE ONLY (see oracle/pyoracle.py).

A line-by-line restatement of the write path of the reference's storage engine, so that the multi-page images the GPU
reader (fbk_batch_upload_rbf) is tested on are the pages the reference itself would have written for the same sequence
of calls — RLE and BitmapPtr leaf cells, leaf splits, branch pages, a root that turned into a branch, root-record pages
chained through their overflow pointer — and not pages laid out by this repository's own idea of the format
(oracle/pyrbf.write_db packs greedily; it stays as the quick fixture builder).  Go is not installed here, so the
reference cannot write these files itself; this is the ceiling of what can be pinned without it.

What is restated, function by function (all in /root/reference/rbf):
    DB.init / initMetaPage / initRootRecordPage / initFreelistPage     db.go:566-604
    Tx.createBitmap, Tx.writeRootRecordPages, writeRootRecords,
        WriteRootRecord                                                tx.go:310-353, 518-573; rbf.go:166-181, 264-296
    Tx.allocatePgno / allocateNewPgno / freePgno, truncateFreelist     tx.go:1150-1216, 155-208
    Tx.Commit's flush: the WAL id the meta page ends up with           tx.go:1957-2035
    Cursor.Seek                                                        cursor.go:1060-1116
    Cursor.Add (new key / array cell)                                  cursor.go:132-165
    Cursor.putLeafCell (fast path and the deserialising slow path),
        putLeafCellFast                                                cursor.go:384-563, 567-636
    splitLeafCells, splitBranchCells, writeRoot, putBranchCells        cursor.go:914-973, 898-910, 694-773
    ConvertToLeafArgs, Cursor.merge, Cursor.AddRoaring                 cursor.go:1299-1408
    page / cell layout: writeLeafCell, writeBranchCell, dataOffset,
        leafPageSize, leafCellsPageSize, branchCellsPageSize           rbf.go:189-216, 527-593, 603-640

ONE simplification, stated: the freelist — in the reference an RBF bitmap of its own (root leaf = page 2) edited through
the same cursor — is kept as a sorted set and rendered at the end as the one leaf page of array cells its final content
gives (putLeafCellFast / deleteLeafCell always leave a page's cells packed in key order, so the bytes of a single-leaf
freelist depend on its content only).  allocatePgno takes the smallest free page number exactly as Cursor.First +
firstValue does.  The model asserts the freelist never outgrows that one leaf page.

Pinned (tests/test_oracle_rbf.py): create "x" + Add(100) + Commit reproduces the database file the reference ships
(ctl/testdata/ok/data: meta, root-record, freelist and leaf page, byte for byte including the WAL id); the cursor tests'
API-level expectations (rbf/cursor_test.go:290-440, 601-640) hold on what the oracle READER gets back from the images."""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import pyrbf as R

PAGE = R.PAGE
T_NONE, T_ARRAY, T_RLE, T_BITMAP, T_BITMAP_PTR = 0, 1, 2, 3, 4  # ContainerType, rbf.go:64-71
LEAF_CELL_HEADER = 8 + 4 + 6   # leafCellHeaderSize rbf.go:90
LEAF_PAGE_HEADER = 4 + 4 + 2   # leafPageHeaderSize :91
BRANCH_CELL = 8 + 4 + 4        # branchCellSize :94
ROOT_REC_PAGE_HEADER = 12      # rootRecordPageHeaderSize :88
FILL = 0.60                    # globalBranchFillPct cursor.go:944


def align8(x: int) -> int:  # rbf.go:298-303
    return x if x % 8 == 0 else x + (8 - (x & 7))


def data_offset(n: int) -> int:  # rbf.go:214-216
    return align8(10 + n * 2)


class LeafCell:  # rbf.go:305-317
    __slots__ = ("key", "typ", "elem_n", "bit_n", "data")

    def __init__(self, key: int, typ: int, elem_n: int, bit_n: int, data: bytes):
        self.key, self.typ, self.elem_n, self.bit_n, self.data = key, typ, elem_n, bit_n, data

    def size(self) -> int:  # leafCell.Size :320-322
        return LEAF_CELL_HEADER + len(self.data)

    def copy(self) -> "LeafCell":
        return LeafCell(self.key, self.typ, self.elem_n, self.bit_n, self.data)


# ---- page (de)serialisation ----------------------------------------------------------------------------------------
def read_cell_n(page: bytes) -> int:
    return struct.unpack_from(">H", page, 8)[0]


def read_flags(page: bytes) -> int:
    return struct.unpack_from(">I", page, 4)[0]


def read_cell_offset(page: bytes, i: int) -> int:
    return struct.unpack_from(">H", page, 10 + 2 * i)[0]


def read_leaf_cell(page: bytes, i: int) -> LeafCell:  # readLeafCell rbf.go:488-512
    o = read_cell_offset(page, i)
    key, typ, elem_n, bit_n = struct.unpack_from("<QIHI", page, o)
    if typ == T_ARRAY:
        data = page[o + 18 : o + 18 + elem_n * 2]
    elif typ == T_RLE:
        data = page[o + 18 : o + 18 + elem_n * 4]
    elif typ == T_BITMAP_PTR:
        data = page[o + 18 : o + 22]
    else:
        data = b""
    return LeafCell(key, typ, elem_n, bit_n, bytes(data))


def read_leaf_cells(page: bytes) -> List[LeafCell]:
    return [read_leaf_cell(page, i) for i in range(read_cell_n(page))]


def leaf_page_size(page: bytes) -> int:  # leafPageSize rbf.go:561-570
    n = read_cell_n(page)
    if n == 0:
        return LEAF_PAGE_HEADER
    return read_cell_offset(page, n - 1) + read_leaf_cell(page, n - 1).size()


def leaf_cells_page_size(cells: List[LeafCell]) -> int:  # leafCellsPageSize :573-579
    return data_offset(len(cells)) + sum(align8(c.size()) for c in cells)


def render_leaf(pgno: int, cells: List[LeafCell]) -> bytes:
    """the page every leaf writer of the reference produces for `cells`: header, index, cells at 8-byte aligned offsets
    from dataOffset(n) on (putLeafCell's group loop cursor.go:512-528, putLeafCellFast :567-636, deleteLeafCell :668-682)"""
    page = bytearray(PAGE)
    struct.pack_into(">IIH", page, 0, pgno, R.LEAF, len(cells))
    off = data_offset(len(cells))
    for j, c in enumerate(cells):
        struct.pack_into(">H", page, 10 + 2 * j, off)  # writeLeafCell rbf.go:585-593
        struct.pack_into("<QIHI", page, off, c.key, c.typ, c.elem_n & 0xFFFF, c.bit_n)
        assert off + LEAF_CELL_HEADER + len(c.data) <= PAGE, "leaf cell write extends beyond page"
        page[off + 18 : off + 18 + len(c.data)] = c.data
        off += align8(c.size())
    return bytes(page)


def read_branch_cells(page: bytes) -> List[Tuple[int, int, int]]:  # readBranchCells rbf.go:623-630 -> (leftKey, flags, childPgno)
    out = []
    for i in range(read_cell_n(page)):
        out.append(struct.unpack_from("<QII", page, read_cell_offset(page, i)))
    return out


def render_branch(pgno: int, cells: List[Tuple[int, int, int]]) -> bytes:  # writeRoot cursor.go:898-910, putBranchCells :737-752
    page = bytearray(PAGE)
    struct.pack_into(">IIH", page, 0, pgno, R.BRANCH, len(cells))
    off = data_offset(len(cells))
    for j, (k, fl, child) in enumerate(cells):
        struct.pack_into(">H", page, 10 + 2 * j, off)  # writeBranchCell rbf.go:634-639
        struct.pack_into("<QII", page, off, k, fl, child)
        off += align8(BRANCH_CELL)
    return bytes(page)


def branch_cells_page_size(cells) -> int:  # rbf.go:603-609
    return data_offset(len(cells)) + len(cells) * align8(BRANCH_CELL)


def split_leaf_cells(cells: List[LeafCell]) -> List[List[LeafCell]]:  # cursor.go:914-942
    slices: List[List[LeafCell]] = [[]]
    data_size = 0
    for cell in cells:
        assert cell.typ != T_BITMAP, "all ContainerTypeBitmap should be ContainerTypeBitmapPtr by now"
        cell_n = len(slices[-1])
        sz = align8(LEAF_CELL_HEADER + len(cell.data))
        thresh = int(float(PAGE) * FILL)
        if cell_n != 0 and (data_offset(cell_n + 1) + data_size + sz) > thresh:
            slices.append([])
            data_size = 0
        elif cell_n != 0 and cell.typ == T_ARRAY and cell.elem_n > R.ARRAY_MAX:
            slices.append([])
            data_size = 0
            sz = PAGE
        slices[-1].append(cell)
        data_size += sz
    return slices


def split_branch_cells(cells):  # cursor.go:948-973
    slices = [[]]
    data_size = 0
    for cell in cells:
        cell_n = len(slices[-1])
        sz = align8(BRANCH_CELL)
        thresh = int(float(PAGE) * FILL)
        if cell_n != 0 and (data_offset(cell_n + 1) + data_size + sz) > thresh:
            slices.append([])
            data_size = 0
        slices[-1].append(cell)
        data_size += sz
    return slices


def search(n: int, f) -> Tuple[int, bool]:  # rbf.go:645-659
    i, j = 0, n
    while i < j:
        h = (i + j) >> 1
        c = f(h)
        if c == 0:
            return h, True
        if c > 0:
            i = h + 1
        else:
            j = h
    return i, False


def convert_to_leaf_args(key: int, c) -> LeafCell:
    """ConvertToLeafArgs (cursor.go:1299-1341) of an oracle container"""
    from . import pyoracle as O

    if c is None or not c.p or c.n == 0:
        return LeafCell(key, T_NONE, 0, 0, b"")
    if c.typ == O.ARRAY:
        if c.length > R.ARRAY_MAX:  # :1309-1315
            return LeafCell(key, T_BITMAP, 0, c.n, np.ascontiguousarray(c.words(), dtype="<u8").tobytes())
        return LeafCell(key, T_ARRAY, c.n, c.n, np.ascontiguousarray(c.data(), dtype="<u2").tobytes())
    if c.typ == O.BITMAP:
        return LeafCell(key, T_BITMAP, 0, c.n, np.ascontiguousarray(c.words(), dtype="<u8").tobytes())
    if c.length > R.RLE_MAX:  # :1327-1334
        return LeafCell(key, T_BITMAP, 0, c.n, np.ascontiguousarray(c.words(), dtype="<u8").tobytes())
    return LeafCell(key, T_RLE, c.length, c.n, np.ascontiguousarray(c.data(), dtype="<u2").tobytes())


class RbfDb:
    """One RBF database file as a sequence of write transactions leaves it (after Commit and checkpoint: every page at
    pgno * 8192 of the data file)."""

    def __init__(self):
        self.pages: Dict[int, bytes] = {}
        self.page_n = 3                       # initMetaPage db.go:578-586
        self.root_record_pgno = 1
        self.freelist_pgno = 2
        self.wal_id = 0
        self.pages[1] = self._header_only(1, R.ROOT_RECORD)   # initRootRecordPage :589-595
        self.free: List[int] = []             # the freelist bitmap's content (see the module docstring), ascending
        self.records: Dict[str, int] = {}     # Tx.rootRecords
        self._dirty: set = set()
        self._dirty_bitmap: set = set()
        self._base: Dict[int, bytes] = dict(self.pages)  # the pages as of the last Commit (what the data file holds)

    @staticmethod
    def _header_only(pgno: int, flags: int) -> bytes:
        page = bytearray(PAGE)
        struct.pack_into(">II", page, 0, pgno, flags)
        return bytes(page)

    # ---- Tx page plumbing ----
    def write_page(self, page: bytes) -> None:  # Tx.writePage tx.go:1280-1283
        pgno = struct.unpack_from(">I", page, 0)[0]
        self.pages[pgno] = page
        self._dirty.add(pgno)

    def write_bitmap_page(self, pgno: int, raw: bytes) -> None:  # Tx.writeBitmapPage :1285-1288
        assert len(raw) == PAGE
        self.pages[pgno] = raw
        self._dirty_bitmap.add(pgno)

    def read_page(self, pgno: int) -> bytes:
        return self.pages[pgno]

    def allocate_pgno(self) -> int:  # Tx.allocatePgno :1150-1193
        if self.free:  # Cursor.First on the freelist, firstValue of its first cell: the smallest free page
            self._dirty.add(self.freelist_pgno)  # c.Remove rewrites the freelist's leaf page
            return self.free.pop(0)
        pgno = self.page_n  # allocateNewPgno :1188-1193
        self.page_n += 1
        return pgno

    def free_pgno(self, pgno: int) -> None:  # Tx.freePgno :1196-1216
        self._dirty.discard(pgno)  # delete(tx.dirtyPages, pgno): what this transaction wrote to the page is never flushed,
        self._dirty_bitmap.discard(pgno)  # the file keeps what the last Commit left there
        if pgno in self._base:
            self.pages[pgno] = self._base[pgno]
        else:
            self.pages.pop(pgno, None)
        assert pgno not in self.free, f"double free: {pgno}"
        self.free.append(pgno)
        self.free.sort()
        self._dirty.add(self.freelist_pgno)  # c.Add on the freelist cursor: putLeafCell on its leaf page

    # ---- root records ----
    def create_bitmap(self, name: str) -> None:  # Tx.createBitmap :310-353
        assert name and name not in self.records
        pgno = self.allocate_pgno()
        self.write_page(render_leaf(pgno, []))
        self.records[name] = pgno
        self.write_root_record_pages()

    def write_root_record_pages(self) -> None:  # Tx.writeRootRecordPages :518-573
        pgno = self.root_record_pgno
        while pgno != 0:  # release all existing root record pages
            page = self.read_page(pgno)
            self.free_pgno(pgno)
            pgno = struct.unpack_from(">I", page, 8)[0]  # WalkRootRecordPages rbf.go:152
        if not self.records:
            self.root_record_pgno = 0
            return
        pgno = self.allocate_pgno()
        self.root_record_pgno = pgno
        names = sorted(self.records)  # immutable.SortedMap iterates in key order
        i = 0
        while i < len(names):
            page = bytearray(PAGE)
            struct.pack_into(">II", page, 0, pgno, R.ROOT_RECORD)
            pos = ROOT_REC_PAGE_HEADER  # writeRootRecords rbf.go:166-181
            short = False
            while i < len(names):
                nb = names[i].encode()
                if PAGE - pos < 6 + len(nb):  # WriteRootRecord: io.ErrShortBuffer rbf.go:276-278
                    short = True
                    break
                struct.pack_into(">IH", page, pos, self.records[names[i]], len(nb))
                page[pos + 6 : pos + 6 + len(nb)] = nb
                pos += 6 + len(nb)
                i += 1
            if short:
                pgno = self.allocate_pgno()
                struct.pack_into(">I", page, 8, pgno)  # writeRootRecordOverflowPgno rbf.go:154-156
            self.write_page(bytes(page))

    # ---- cursor ----
    def seek(self, root: int, key: int):
        """Cursor.Seek (cursor.go:1060-1116) -> (stack [(pgno, index)], exact)"""
        stack = []
        pgno = root
        while True:
            buf = self.read_page(pgno)
            typ = read_flags(buf)
            n = read_cell_n(buf)
            if typ == R.BRANCH:
                cells = read_branch_cells(buf)
                index, xact = search(n, lambda i: 0 if key == cells[i][0] else (-1 if key < cells[i][0] else 1))
                if not xact and index > 0:
                    index -= 1
                stack.append((pgno, index))
                pgno = cells[index][2]
            elif typ == R.LEAF:
                keys = [read_leaf_cell(buf, i).key for i in range(n)]
                index, xact = search(n, lambda i: 0 if key == keys[i] else (-1 if key < keys[i] else 1))
                stack.append((pgno, index))
                return stack, xact
            else:
                raise ValueError(f"rbf.Cursor.Seek(): invalid page type: pgno={pgno} type={typ}")

    def put_leaf_cell(self, stack, cell_in: LeafCell) -> None:
        """Cursor.putLeafCell (cursor.go:384-563)"""
        pgno, index = stack[-1]
        leaf_page = self.read_page(pgno)
        cell_n = read_cell_n(leaf_page)
        is_insert = index >= cell_n or read_leaf_cell(leaf_page, index).key != cell_in.key
        new_est = leaf_page_size(leaf_page)
        if is_insert:
            new_est += cell_in.size() + 2  # leafCellIndexElemSize
        else:
            new_est += cell_in.size() - read_leaf_cell(leaf_page, index).size()
        use_fast = new_est + 16 <= PAGE
        if use_fast and not is_insert and read_leaf_cell(leaf_page, index).typ != cell_in.typ:
            use_fast = False
        cells = read_leaf_cells(leaf_page)
        if use_fast:
            # putLeafCellFast (:567-636) shifts the bytes before and after the cell around a freshly written one; with
            # every page packed from dataOffset(n) on at 8-byte aligned offsets that is the packed page of the new list
            if is_insert:
                cells.insert(index, cell_in)
            else:
                cells[index] = cell_in
            self.write_page(render_leaf(pgno, cells))
            return
        inn = cell_in.copy()
        cell = inn.copy()
        if is_insert:
            if inn.typ == T_BITMAP:  # :426-437
                bitmap_pgno = self.allocate_pgno()
                cell = LeafCell(inn.key, T_BITMAP_PTR, inn.elem_n, inn.bit_n, struct.pack("<I", bitmap_pgno))
            cells.insert(index, LeafCell(0, 0, 0, 0, b""))  # :439-440
        else:
            prev = cells[index]  # FB-1239 :444-458
            if prev.typ == T_BITMAP_PTR:
                if inn.typ == T_BITMAP_PTR:
                    if inn.data != prev.data:
                        self.free_pgno(struct.unpack("<I", prev.data)[0])
                elif inn.typ != T_BITMAP:
                    self.free_pgno(struct.unpack("<I", prev.data)[0])
            if inn.typ == T_BITMAP:  # :460-473
                cell = cells[index].copy()
                if cell.typ != T_BITMAP_PTR:
                    bitmap_pgno = self.allocate_pgno()
                    cell.typ = T_BITMAP_PTR
                    cell.data = struct.pack("<I", bitmap_pgno)
                    cell.elem_n = inn.elem_n
                cell.bit_n = inn.bit_n
        if inn.typ == T_ARRAY and inn.elem_n > R.ARRAY_MAX:  # :476-490
            a = np.zeros(PAGE // 8, dtype=np.uint64)
            for v in np.frombuffer(inn.data, dtype="<u2"):
                a[int(v) // 64] |= np.uint64(1) << np.uint64(int(v) % 64)
            inn.typ = T_BITMAP
            inn.data = a.astype("<u8").tobytes()
            cell.typ = T_BITMAP_PTR
            cell.data = struct.pack("<I", self.allocate_pgno())
        cells[index] = cell
        groups = [cells]
        if leaf_cells_page_size(cells) >= PAGE:  # :495-499
            groups = split_leaf_cells(cells)
        new_root = len(groups) > 1 and len(stack) == 1  # :502
        parents = []
        for i, group in enumerate(groups):
            if i == 0 and not new_root:
                child = pgno
            else:
                child = self.allocate_pgno()
            if inn.typ == T_BITMAP:  # :519-525 (inside the loop in the reference too: the same page every time)
                self.write_bitmap_page(struct.unpack("<I", cell.data)[0], inn.data)
            self.write_page(render_leaf(child, group))
            parents.append((group[0].key, 0, child))
        if len(groups) == 1:
            return
        if len(stack) == 1:  # :553-556
            self.write_page(render_branch(pgno, parents))  # writeRoot(origPgno, parents)
            return
        self.put_branch_cells(stack, len(stack) - 2, parents)

    def put_branch_cells(self, stack, stack_index: int, new_cells) -> None:
        """Cursor.putBranchCells (cursor.go:694-773)"""
        pgno, index = stack[stack_index]
        page = self.read_page(pgno)
        cells = read_branch_cells(page)
        if not cells:
            cells = [(0, 0, 0)]
        cells[index] = new_cells[0]
        if len(new_cells) > 1:
            cells[index + 1 : index + 1] = list(new_cells[1:])
        groups = [cells]
        if branch_cells_page_size(cells) > PAGE:
            groups = split_branch_cells(cells)
        parents = []
        orig = struct.unpack_from(">I", page, 0)[0]
        new_root = len(groups) > 1 and stack_index == 0
        for i, group in enumerate(groups):
            child = orig if (i == 0 and not new_root) else self.allocate_pgno()
            parents.append((group[0][0], 0, child))
            self.write_page(render_branch(child, group))
        if len(groups) == 1:
            return
        if stack_index == 0:
            self.write_page(render_branch(orig, parents))  # writeRoot
            return
        self.put_branch_cells(stack, stack_index - 1, parents)

    def add(self, name: str, v: int) -> bool:
        """Cursor.Add (cursor.go:132-165) for a value under a new key or in an array cell below ArrayMaxSize — what the
        reference's own fixture files were written with; other cell types go through add_roaring"""
        root = self.records[name]
        hi, lo = v >> 16, v & 0xFFFF
        stack, exact = self.seek(root, hi)
        if not exact:
            self.put_leaf_cell(stack, LeafCell(hi, T_ARRAY, 1, 1, struct.pack("<H", lo)))
            return True
        cell = read_leaf_cell(self.read_page(stack[-1][0]), stack[-1][1])
        assert cell.typ == T_ARRAY, "add(): only array cells (use add_roaring)"
        a = list(np.frombuffer(cell.data, dtype="<u2"))
        if lo in a:
            return False
        a = sorted(a + [lo])
        self.put_leaf_cell(stack, LeafCell(cell.key, T_ARRAY, len(a), cell.bit_n + 1, np.asarray(a, dtype="<u2").tobytes()))
        return True

    def add_roaring(self, name: str, items) -> bool:
        """Cursor.AddRoaring (cursor.go:1376-1408) with Cursor.merge (:1343-1374); items: [(key, oracle container)] in key order"""
        from . import pyoracle as O

        root = self.records[name]
        changed = False
        for hi, cont in items:
            leaf = convert_to_leaf_args(hi, cont)
            if leaf.bit_n == 0:
                continue
            stack, exact = self.seek(root, hi)
            if not exact:
                self.put_leaf_cell(stack, leaf)
                changed = True
                continue
            cell = read_leaf_cell(self.read_page(stack[-1][0]), stack[-1][1])
            if cell.typ == T_ARRAY:
                stored = O.OContainer.array(np.frombuffer(cell.data, dtype="<u2"))
            elif cell.typ == T_BITMAP_PTR:
                raw = self.read_page(struct.unpack("<I", cell.data)[0])
                stored = O.OContainer.bitmap(np.frombuffer(raw, dtype="<u8").copy())
            else:
                stored = O.OContainer.run([tuple(x) for x in np.frombuffer(cell.data, dtype="<u2").reshape(-1, 2).tolist()])
            res = O.optimize(O.union(cont, stored))  # roaring.Union = union + optimize (roaring.go:7610-7615)
            if res.n != cont.n:  # `res.N() != data.N()`: compared with the INCOMING container (:1367)
                self.put_leaf_cell(stack, convert_to_leaf_args(hi, res))
                changed = True
        return changed

    # ---- commit ----
    def commit(self) -> None:
        """Tx.Commit (tx.go:109-150): truncateFreelist, then flush — every dirty page, every dirty bitmap page behind a
        header page, then the meta page, each one WAL id (:1957-2035)"""
        while self.free and self.free[-1] >= self.page_n - 1:  # truncateLastFreePage :168-208
            self.free.pop()
            self._dirty.add(self.freelist_pgno)
            self.page_n -= 1
        if self._dirty or self._dirty_bitmap:
            self.wal_id += len(self._dirty) + 2 * len(self._dirty_bitmap) + 1
        self._dirty, self._dirty_bitmap = set(), set()
        self._base = dict(self.pages)

    def _freelist_page(self) -> bytes:
        cells = []
        for hi in sorted({p >> 16 for p in self.free}):
            vals = [p & 0xFFFF for p in self.free if p >> 16 == hi]
            cells.append(LeafCell(hi, T_ARRAY, len(vals), len(vals), np.asarray(vals, dtype="<u2").tobytes()))
        assert leaf_cells_page_size(cells) < PAGE, "the freelist model holds one leaf page"
        return render_leaf(self.freelist_pgno, cells)

    def image(self) -> bytes:
        """the data file after Commit + checkpoint"""
        meta = bytearray(PAGE)
        meta[:4] = R.MAGIC
        struct.pack_into(">I", meta, 8, self.page_n)
        struct.pack_into(">q", meta, 12, self.wal_id)
        struct.pack_into(">II", meta, 20, self.root_record_pgno, self.freelist_pgno)
        out = [bytes(meta)]
        for pgno in range(1, self.page_n):
            if pgno == self.freelist_pgno:
                out.append(self._freelist_page())
            else:
                out.append(self.pages.get(pgno, bytes(PAGE)))
        return b"".join(out)
