#!/usr/bin/env python3
"""Record EVERY page of the four RBF database files the reference ships, as the non-zero prefix of
each page (data only):

  rbf/testdata/check/bad-freelist/data   rbf/tx_test.go:1277-1291   meta, root records, a page 2 with
                                         the BRANCH flag where the freelist leaf belongs, leaf {key 0: [100]}
  rbf/testdata/check/bad-bitmap/data     rbf/tx_test.go:1293-1303   page 3 rewritten as a BRANCH page whose
                                         one cell points at page 65537
  ctl/testdata/ok/data                   ctl/rbf_check_test.go:16-28, rbf_dump_test.go:16-28  a consistent file
  ctl/testdata/err-invalid-page-type/data  ctl/rbf_check_test.go:30-42  the same + an all-zero page 4

These are all the reference-written page images in the tree.  Page kinds they contain: meta page,
root-record page, an (empty) freelist leaf, one leaf page with ONE array cell, one branch page with
ONE cell (its child out of bounds).  Not contained anywhere: RLE cells, BitmapPtr cells + bitmap
pages, branch pages that lead somewhere, multi-page root records — those stay pinned only by the
oracle's own writer (oracle/pyrbf.py) following rbf/rbf.go's documented layout.

    python tests/golden/extract_rbf_fixture.py [/root/reference]
"""
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
FILES = {
    "bad-freelist": "rbf/testdata/check/bad-freelist/data",
    "bad-bitmap": "rbf/testdata/check/bad-bitmap/data",
    "ok": "ctl/testdata/ok/data",
    "err-invalid-page-type": "ctl/testdata/err-invalid-page-type/data",
}
ONE_ARRAY = {"root_pgno": 3, "containers": [{"key": 0, "type": "array", "n": 1, "values": [100]}]}


def pages_of(path):
    raw = open(os.path.join(REF, path), "rb").read()
    assert len(raw) % 8192 == 0
    return [raw[p * 8192: (p + 1) * 8192].rstrip(b"\0").hex() for p in range(len(raw) // 8192)]


def main():
    files = {name: {"path": path, "pages_hex_prefix": pages_of(path)} for name, path in FILES.items()}
    files["bad-freelist"]["expect"] = ONE_ARRAY
    files["ok"]["expect"] = ONE_ARRAY
    files["err-invalid-page-type"]["expect"] = ONE_ARRAY  # the stray page 4 is not reachable from bitmap "x"
    files["bad-bitmap"]["error"] = "cannot read page: pgno=65537 parent=3 err=rbf: page read out of bounds: pgno=65537 max=3 (rbf/tx_test.go:1301)"
    with open(os.path.join(OUT, "rbf_fixture.json"), "w") as f:
        json.dump({"source": "rbf/testdata/check/*, ctl/testdata/{ok,err-invalid-page-type} (rbf/tx_test.go:1277-1303, ctl/rbf_check_test.go:16-42)",
                   "page_size": 8192, "bitmap": "x", "files": files,
                   # kept for the round-1 tests: the bad-freelist file and the bad-bitmap pages under their old names
                   "pages_hex_prefix": files["bad-freelist"]["pages_hex_prefix"],
                   "bad_bitmap_pages_hex_prefix": files["bad-bitmap"]["pages_hex_prefix"],
                   "bad_bitmap_error": files["bad-bitmap"]["error"], "expect": ONE_ARRAY}, f, indent=1)
    for n, d in files.items():
        print(n, [len(p) // 2 for p in d["pages_hex_prefix"]])


if __name__ == "__main__":
    main()
