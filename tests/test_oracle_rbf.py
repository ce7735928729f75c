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
