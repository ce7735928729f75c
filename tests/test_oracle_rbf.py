"""Pin the RBF page reader/writer of the oracle (oracle/pyrbf.py) to the database file the
reference ships (rbf/testdata/check/bad-freelist, tests/golden/rbf_fixture.json) and check the
writer against the reader on b-trees with branch pages, bitmap pages and all cell types.
CPU only."""
import json
import os
import struct

import numpy as np

import datagen as D

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "rbf_fixture.json")))


def fixture_file() -> bytes:
    return b"".join(bytes.fromhex(h).ljust(FIX["page_size"], b"\0") for h in FIX["pages_hex_prefix"])


def file_image(name: str) -> bytes:
    """one of the four reference-written database files (tests/golden/rbf_fixture.json "files")"""
    return b"".join(bytes.fromhex(h).ljust(FIX["page_size"], b"\0") for h in FIX["files"][name]["pages_hex_prefix"])


def test_every_reference_written_file():
    """All four RBF files the reference ships (rbf/testdata/check/*, ctl/testdata/{ok,
    err-invalid-page-type}): the oracle's reader finds bitmap "x" and its one array cell in the three
    readable ones and fails on bad-bitmap's branch cell exactly where the reference does
    (pgno=65537, rbf/tx_test.go:1301)."""
    import pytest

    from oracle import pyrbf

    for name, d in FIX["files"].items():
        f = file_image(name)
        assert len(f) == 8192 * len(d["pages_hex_prefix"]) and f[:4] == b"\xffRBF"
        root = pyrbf.find_root(f, FIX["bitmap"])
        assert root == 3
        if "expect" in d:
            conts = pyrbf.read_bitmap(f, root)
            assert [(k, t, n, p.tolist()) for k, t, n, p in conts] == [(0, 1, 1, [100])], name
        else:
            # page 3 carries the BRANCH flag and one branch cell {leftKey 0, flags, childPgno 65537}
            assert struct.unpack_from(">IIH", f, 3 * 8192) == (3, pyrbf.BRANCH, 1)
            with pytest.raises(Exception):
                pyrbf.read_bitmap(f, root)


def random_fragment(rng, n_rows, oracle):
    """containers of a fragment: key = row*16 + slot, RBF-legal encodings"""
    out = []
    for r in range(n_rows):
        row = D.random_row(rng, 0)
        for k, c in sorted(row.items()):
            if c.n == 0:
                continue
            if (c.typ == 1 and c.n > 4079) or (c.typ == 3 and c.length > 2039):
                c = oracle.OContainer.bitmap(c.words())  # RBF stores these as bitmap pages (rbf.go:37-42)
            payload = c.data() if c.typ != 2 else c.words()
            out.append(((r * 5 + 2) * 16 + (k & 15), c.typ, c.n, np.asarray(payload)))
    return out


def test_reference_written_file():
    from oracle import pyrbf

    f = fixture_file()
    root = pyrbf.find_root(f, FIX["bitmap"])
    assert root == FIX["expect"]["root_pgno"]
    conts = pyrbf.read_bitmap(f, root)
    assert [(k, t, n, p.tolist()) for k, t, n, p in conts] == [(0, 1, 1, [100])]
    # the header fields the writer must reproduce: big-endian page header, cell at align8(10 + 2)
    page = f[3 * 8192 : 4 * 8192]
    assert struct.unpack_from(">IIH", page, 0) == (3, 2, 1) and struct.unpack_from(">H", page, 10)[0] == 16
    assert pyrbf.write_db({"x": conts})[2 * 8192 : 3 * 8192][:35] == page[:35].replace(b"\0\0\0\3", b"\0\0\0\2", 1)


def test_writer_reader_round_trip(oracle):
    from oracle import pyrbf

    rng = D.rng_for(97)
    frag = random_fragment(rng, 40, oracle)
    other = random_fragment(rng, 3, oracle)
    f = pyrbf.write_db({"i/f/standard/0": frag, "i/g/standard/7": other}, leaf_cells_per_page=5, branch_fanout=4)
    assert len(f) % 8192 == 0 and f[:4] == b"\xffRBF"
    for name, conts in (("i/f/standard/0", frag), ("i/g/standard/7", other)):
        root = pyrbf.find_root(f, name)
        back = pyrbf.read_bitmap(f, root)
        assert len(back) == len(conts)
        for (k0, t0, n0, p0), (k1, t1, n1, p1) in zip(conts, back):
            assert (k0, t0, n0) == (k1, t1, n1) and np.array_equal(np.asarray(p0).reshape(-1), p1.reshape(-1))
    # the 40-row tree needs branch pages above the leaves at this fan-out
    root = pyrbf.find_root(f, "i/f/standard/0")
    assert struct.unpack_from(">I", f, root * 8192 + 4)[0] == pyrbf.BRANCH


# ---- the restated WRITER (oracle/pyrbf_writer.py) -------------------------------------------------------------------
def test_writer_restatement_reproduces_the_reference_written_file():
    """createBitmap("x") + Add(100) + Commit through the restated write path = the database file the reference ships
    (ctl/testdata/ok/data): meta page (page count 4, WAL id 4, root-record page 1, freelist page 2), root-record page
    {3, "x"}, the empty freelist leaf and the leaf page with the one array cell — every byte the fixture holds."""
    from oracle import pyrbf_writer as W

    db = W.RbfDb()
    db.create_bitmap("x")
    assert db.add("x", 100) and not db.add("x", 100)
    db.commit()
    img = db.image()
    ref = file_image("ok")
    assert len(img) == len(ref) == 4 * 8192
    for pg, h in enumerate(FIX["files"]["ok"]["pages_hex_prefix"]):
        want = bytes.fromhex(h)
        assert img[pg * 8192 : pg * 8192 + len(want)] == want, pg
        assert img[pg * 8192 + len(want) : (pg + 1) * 8192] == bytes(8192 - len(want)), pg  # (the fixture keeps each page up to its last non-zero byte)


def test_writer_restatement_add_roaring_sequences(oracle):
    """Cursor.AddRoaring through the restated writer, read back with the oracle READER.  Cursor.merge compares the union
    with the INCOMING container (`res.N() != data.N()`, cursor.go:1367): an incoming SUPERSET of what is stored is reported
    "unchanged" and NOT written (the reference's behaviour, restated as it is), an incoming subset rewrites the cell."""
    from oracle import pyrbf, pyrbf_writer as W

    O = oracle

    def cells(db):
        img = db.image()
        return [(k, t, n, p.tolist()) for k, t, n, p in pyrbf.read_bitmap(img, pyrbf.find_root(img, "x"))]

    db = W.RbfDb()
    db.create_bitmap("x")
    assert db.add_roaring("x", [(0, O.OContainer.array([1, 2]))])
    assert not db.add_roaring("x", [(0, O.OContainer.array([1, 2, 3]))])   # union N = 3 = incoming N: not written
    assert cells(db) == [(0, 1, 2, [1, 2])]
    assert db.add_roaring("x", [(0, O.OContainer.array([7]))])             # union N = 3 != 1: written
    assert cells(db) == [(0, 1, 3, [1, 2, 7])]
    assert db.add_roaring("x", [(0, O.OContainer.array([2, 3, 4, 5, 6]))])  # {1..7}: roaring.Union optimize()s into one run
    assert cells(db) == [(0, 3, 7, [[1, 7]])]
    # a bitmap container: BitmapPtr cell + a raw bitmap page; an array merged into it: still a bitmap page, more bits
    db = W.RbfDb()
    db.create_bitmap("x")
    assert db.add_roaring("x", [(0, O.OContainer.bitmap(D.words_of(np.arange(0, 65536, 2))))])
    page_n = db.page_n
    assert db.add_roaring("x", [(0, O.OContainer.array([1, 3, 5]))])
    assert db.page_n == page_n  # the bitmap page is rewritten in place (cursor.go:460-473)
    db.commit()
    img = db.image()
    (k, t, n, p), = pyrbf.read_bitmap(img, pyrbf.find_root(img, "x"))
    assert (k, t, n) == (0, 2, 32771) and (p == D.words_of(np.union1d(np.arange(0, 65536, 2), [1, 3, 5]))).all()
    # ... and a run that swallows it: the cell becomes RLE and the bitmap page goes to the freelist (FB-1239, cursor.go:444-458)
    assert db.add_roaring("x", [(0, O.OContainer.run([(0, 65000)]))])
    assert db.free == [page_n - 1]
    db.commit()
    assert db.page_n == page_n - 1 and db.free == []  # Commit truncates a free page at the end of the file (tx.go:155-208)
    assert cells(db) == [(0, 3, 65268, [[0, 65000]] + [[v, v] for v in range(65002, 65536, 2)])]
    # RLE conversion at the cell limit (TestCursor_RLEConversion :601-640): 2039 runs stay RLE, 2040 become a bitmap page
    for nr, want_t in ((2039, 3), (2040, 2)):
        db = W.RbfDb()
        db.create_bitmap("x")
        assert db.add_roaring("x", [(7, O.OContainer.run([(16 * i, 16 * i + 9) for i in range(nr)]))])
        db.commit()
        img = db.image()
        (k, t, n, p), = pyrbf.read_bitmap(img, pyrbf.find_root(img, "x"))
        assert (k, t, n) == (7, want_t, 10 * nr)
    # array at the cell limit: 4079 values an array cell, 4080 a bitmap page (ArrayMaxSize, rbf.go:37-39)
    for nv, want_t in ((4079, 1), (4080, 2)):
        db = W.RbfDb()
        db.create_bitmap("x")
        assert db.add_roaring("x", [(1, O.OContainer.array(np.arange(nv) * 3))])
        db.commit()
        img = db.image()
        (k, t, n, p), = pyrbf.read_bitmap(img, pyrbf.find_root(img, "x"))
        assert (k, t, n) == (1, want_t, nv)


def build_multi_page_db(oracle, n_names=3):
    """A database image with everything the reader has to get right, written by the restated writer: a fragment of 2600
    containers (arrays, RLE cells, BitmapPtr cells) added in three AddRoaring calls that interleave keys — leaf pages
    split, the root turns into a branch page, the branch itself splits (three levels) — plus enough bitmaps with long
    names that the root records spill into a chained overflow page."""
    from oracle import pyrbf_writer as W

    O = oracle
    rng = D.rng_for(4711)
    db = W.RbfDb()
    names = [f"idx/field-{i:04d}/" + "v" * 150 + f"/standard/{i}" for i in range(n_names)]
    expect = {}
    for i, name in enumerate(names):
        db.create_bitmap(name)
    frag = "i/f/standard/0"
    db.create_bitmap(frag)
    conts = {}
    keys = list(range(0, 5200, 2))
    for rnd in range(3):
        items = []
        for key in keys[rnd::3]:
            kind = int(rng.integers(0, 5))
            if kind == 0:
                c = O.OContainer.array(np.sort(rng.choice(65536, int(rng.integers(1, 40)), replace=False)))
            elif kind == 1:
                c = O.OContainer.array(np.sort(rng.choice(65536, int(rng.integers(300, 1500)), replace=False)))
            elif kind == 2:
                st = np.sort(rng.choice(4000, int(rng.integers(1, 300)), replace=False)) * 16
                c = O.OContainer.run([(int(s), int(s) + int(rng.integers(2, 14))) for s in st])
            elif kind == 3:
                c = O.OContainer.bitmap(D.words_of(np.sort(rng.choice(65536, 20000, replace=False))))
            else:
                c = O.OContainer.run([(0, 65535)])
            items.append((key, c))
            conts[key] = c
        assert db.add_roaring(frag, items)
        db.commit()
    # a second pass merges into a third of the keys (type changes: array -> bitmap page, bitmap page -> full run frees its page)
    items = []
    for key in keys[::3]:
        extra = O.OContainer.run([(0, 65535)]) if key % 4 == 0 else O.OContainer.array(np.sort(rng.choice(65536, 3000, replace=False)))
        items.append((key, extra))
        res = O.optimize(O.union(extra, conts[key]))
        if res.n != extra.n:  # Cursor.merge's rule (cursor.go:1367): an incoming superset is not written
            conts[key] = res
    db.add_roaring(frag, items)
    db.commit()
    # the other bitmaps get one small container each
    for i, name in enumerate(names):
        c = O.OContainer.array([i, i + 7])
        db.add_roaring(name, [(i, c)])
        expect[name] = {i: c}
    db.commit()
    expect[frag] = conts
    return db, expect


def test_writer_restatement_multi_page_image_structure(oracle):
    from oracle import pyrbf, pyrbf_writer as W

    db, expect = build_multi_page_db(oracle, n_names=60)
    img = db.image()
    assert len(img) == db.page_n * 8192
    # root records: chained through the overflow pointer (60 names of ~180 bytes do not fit one page)
    first = struct.unpack_from(">I", img, 20)[0]
    nxt = struct.unpack_from(">I", img, first * 8192 + 8)[0]
    assert nxt != 0 and struct.unpack_from(">I", img, nxt * 8192 + 4)[0] == pyrbf.ROOT_RECORD
    # the fragment's root kept its page number and is a branch whose children are branches (three levels)
    root = pyrbf.find_root(img, "i/f/standard/0")
    assert root == db.records["i/f/standard/0"]
    assert struct.unpack_from(">I", img, root * 8192 + 4)[0] == pyrbf.BRANCH
    kids = W.read_branch_cells(img[root * 8192 : (root + 1) * 8192])
    assert len(kids) >= 2 and all(struct.unpack_from(">I", img, c * 8192 + 4)[0] == pyrbf.BRANCH for _, _, c in kids)
    # every cell type occurs; every page the writer produced respects the page size and the 8-byte alignment of its cells
    types = set()

    def walk(pgno):
        page = img[pgno * 8192 : (pgno + 1) * 8192]
        if W.read_flags(page) == pyrbf.BRANCH:
            cells = W.read_branch_cells(page)
            assert cells and [k for k, _, _ in cells] == sorted(k for k, _, _ in cells)
            for _, _, c in cells:
                walk(c)
            return
        n = W.read_cell_n(page)
        assert n >= 1 and all(W.read_cell_offset(page, i) % 8 == 0 for i in range(n))
        assert W.leaf_page_size(page) <= 8192
        types.update(W.read_leaf_cell(page, i).typ for i in range(n))

    walk(root)
    assert types == {W.T_ARRAY, W.T_RLE, W.T_BITMAP_PTR}
    # and the oracle reader gets back exactly what was put in (ConvertToLeafArgs' encodings of the merged containers)
    for name, conts in expect.items():
        back = pyrbf.read_bitmap(img, pyrbf.find_root(img, name))
        assert [k for k, _, _, _ in back] == sorted(conts)
        for k, t, n, p in back:
            c = conts[k]
            assert n == c.n and (pyrbf.leaf_to_container((k, t, n, p)).words() == c.words()).all(), (name, k)
