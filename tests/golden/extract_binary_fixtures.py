#!/usr/bin/env python3
"""Copy the serialised-fragment files the reference ships into tests/golden/ (binary DATA written by the reference's
own Bitmap.WriteTo, no code) and record their provenance:

  testdata/sample_view/0                      the default of the -fragment flag of the fragment tests
                                              (fragment_internal_test.go:33-34): a Pilosa-format fragment image,
                                              cookie 12348, 14 207 containers, 1000 rows, 35 001 bits
  cmd/roaring-migrate/testdata/data-dir/repository/{_exists,language}/views/standard/fragments/222
                                              two 21-byte fragment files: an empty container table followed by a
                                              one-entry OPS LOG (op layout roaring.go:6325-6431)
  rbf/cursor_test.go                          (name, wantChanged) of TestCursor_AddRoaring's table (:280-430), in order

    python tests/golden/extract_binary_fixtures.py [/root/reference]
"""
import hashlib
import json
import os
import re
import shutil
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    meta = {"files": {}}
    src = os.path.join(REF, "testdata", "sample_view", "0")
    dst = os.path.join(OUT, "sample_view_0.roaring")
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    raw = open(dst, "rb").read()
    meta["files"]["sample_view_0.roaring"] = {"source": "testdata/sample_view/0 (fragment_internal_test.go:33-34)", "bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest()}
    for field in ("_exists", "language"):
        p = os.path.join(REF, "cmd", "roaring-migrate", "testdata", "data-dir", "repository", field, "views", "standard", "fragments", "222")
        raw = open(p, "rb").read()
        meta["files"][f"migrate_{field}_222"] = {"source": os.path.relpath(p, REF), "hex": raw.hex(), "sha256": hashlib.sha256(raw).hexdigest()}
    go = open(os.path.join(REF, "rbf", "cursor_test.go")).read()
    m = re.search(r"^func TestCursor_AddRoaring\(", go, re.M)
    blk = go[m.start(): go.index("func TestCursor_RLETesting", m.start())]
    line0 = go.count("\n", 0, m.start()) + 1
    cases = [{"name": n, "wantChanged": w == "true", "line": line0 + blk.count("\n", 0, q.start())}
             for q in re.finditer(r'name:\s+"([^"]*)",.*?wantChanged:\s+(true|false)', blk, re.S) for n, w in [q.groups()]]
    meta["cursor_add_roaring"] = {"source": f"rbf/cursor_test.go:{line0}", "cases": cases}
    consts = open(os.path.join(REF, "rbf", "rbf.go")).read()
    meta["ArrayMaxSize"] = int(re.search(r"ArrayMaxSize\s*=\s*(\d+)", consts).group(1))
    meta["RLEMaxSize"] = int(re.search(r"RLEMaxSize\s*=\s*(\d+)", consts).group(1))
    json.dump(meta, open(os.path.join(OUT, "binary_fixtures.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in meta.items() if k != "cursor_add_roaring"}, indent=1))
    print([(c["name"], c["wantChanged"]) for c in cases])


if __name__ == "__main__":
    main()
