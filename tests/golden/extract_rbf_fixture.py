#!/usr/bin/env python3
"""Record the RBF database file the reference ships (rbf/testdata/check/bad-freelist/data, used
by rbf/tx_test.go:1277-1291; pages 0, 1 and 3 were written by the reference: a meta page, a
root record page for bitmap "x" and a leaf page with one array cell {key 0: [100]}) as the
non-zero prefix of each page.  Data only.

    python tests/golden/extract_rbf_fixture.py [/root/reference]
"""
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    raw = open(os.path.join(REF, "rbf", "testdata", "check", "bad-freelist", "data"), "rb").read()
    assert len(raw) % 8192 == 0
    pages = []
    for p in range(len(raw) // 8192):
        pg = raw[p * 8192 : (p + 1) * 8192]
        n = len(pg.rstrip(b"\0"))
        pages.append(pg[:n].hex())
    bad = open(os.path.join(REF, "rbf", "testdata", "check", "bad-bitmap", "data"), "rb").read()
    bad_pages = [bad[p * 8192 : (p + 1) * 8192].rstrip(b"\0").hex() for p in range(len(bad) // 8192)]
    with open(os.path.join(OUT, "rbf_fixture.json"), "w") as f:
        json.dump({"source": "rbf/testdata/check/bad-freelist/data (rbf/tx_test.go:1277-1291)", "page_size": 8192, "pages_hex_prefix": pages,
                   "bad_bitmap_pages_hex_prefix": bad_pages,
                   "bad_bitmap_error": "cannot read page: pgno=65537 parent=3 err=rbf: page read out of bounds: pgno=65537 max=3 (rbf/tx_test.go:1301)",
                   "bitmap": "x", "expect": {"root_pgno": 3, "containers": [{"key": 0, "type": "array", "n": 1, "values": [100]}]}}, f, indent=1)
    print("rbf_fixture.json:", [len(p) // 2 for p in pages])


if __name__ == "__main__":
    main()
