#!/usr/bin/env python3
"""Record the reference's serialisation fixtures (TestUnmarshalRoaringWithNoErrors /
WithErrors, roaring/roaring_internal_test.go:3793-3880, and the data file
roaring/testdata/bitmapcontainer.roaringbitmap) as hex strings with their expected counts.
Only data literals are recorded, never code.

    python tests/golden/extract_wire_fixtures.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(os.path.join(REF, "roaring", "roaring_internal_test.go")).read()
    m = re.search(r"^func TestUnmarshalRoaringWithNoErrors\(", src, re.M)
    blk = src[m.start(): src.index("func TestUnmarshalRoaringWithErrors", m.start())]
    line0 = src.count("\n", 0, m.start()) + 1
    ok = []
    for q in re.finditer(r'roaringData:\s+"([0-9A-Fa-f]+)",\s+count:\s+(\d+),\s+expectedBits:\s+"\[([^\]]*)\]"', blk):
        ok.append({"hex": q.group(1), "count": int(q.group(2)), "bits": [int(x) for x in q.group(3).split()], "line": line0 + blk.count("\n", 0, q.start())})
    q = re.search(r'roaringFileName:\s+"([^"]+)",\s+count:\s+(\d+)', blk)
    raw = open(os.path.join(REF, "roaring", q.group(1)), "rb").read()
    ok.append({"hex": raw.hex().upper(), "count": int(q.group(2)), "bits": None, "file": "roaring/" + q.group(1), "line": line0 + blk.count("\n", 0, q.start())})
    m2 = re.search(r"^func TestUnmarshalRoaringWithErrors\(", src, re.M)
    blk2 = src[m2.start(): m2.start() + 1500]
    bad = [{"hex": h, "error": e} for h, e in re.findall(r'hexString:\s+"([0-9A-Fa-f]+)",\s+expectedError:\s+"([^"]*)"', blk2)]
    empty_ok = re.findall(r'hexString:\s+"(3C30[0-9A-Fa-f]+)"', blk2)
    assert len(ok) == 3 and len(bad) == 2 and empty_ok == ["3C30000000000000"]
    with open(os.path.join(OUT, "wire_fixtures.json"), "w") as f:
        json.dump({"source": "roaring/roaring_internal_test.go:3793-3880", "ok": ok, "errors": bad, "pilosa_empty_ok": empty_ok}, f, indent=1)
    print("wire_fixtures.json:", [(len(o["hex"]) // 2, o["count"]) for o in ok], bad, empty_ok)


if __name__ == "__main__":
    main()
