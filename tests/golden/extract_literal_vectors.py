#!/usr/bin/env python3
"""Mechanical extraction of the LITERALS of two reference tests whose data used to be transcribed by hand:

  * TestBitmap_IntersectionCount_{ArrayArray, ArrayRun, RunRun, BitmapRun, ArrayBitmap, BitmapBitmap, Mixed}
    (roaring/roaring_test.go): every roaring.NewFileBitmap(<literal values>) and every expected count (`n != K`)
    in source order.  (The bitmaps these tests build with loops cannot be extracted: tests/golden/go_bitmap_vectors.py
    restates the loops; tests/test_golden_transcriptions.py checks its literals against this extraction.)
  * TestBSIAddCases (bsi_test.go): the positions / a / b slices of every case.
  * TestOpLogWriteUnmarshal (roaring/roaring_internal_test.go): the twelve ops it writes to an ops log and reads back.

    python tests/golden/extract_literal_vectors.py [/root/reference]  ->  tests/golden/literal_vectors.json
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def func_body(src: str, name: str) -> str:
    i = src.index("func " + name + "(")
    j = src.find("\nfunc ", i + 1)
    return src[i:j] if j >= 0 else src[i:]


def ints(s: str):
    return [int(x) for x in re.findall(r"\d+", s)]


out = {"source": {"roaring_test": "roaring/roaring_test.go", "bsi_test": "bsi_test.go"}, "intersection_count": {}, "bsi_add_cases": []}
src = open(os.path.join(REF, "roaring", "roaring_test.go")).read()
for name in ("ArrayArray", "ArrayRun", "RunRun", "BitmapRun", "ArrayBitmap", "BitmapBitmap", "Mixed"):
    body = func_body(src, "TestBitmap_IntersectionCount_" + name)
    out["intersection_count"][name] = {
        "new_file_bitmap_literals": [ints(m) for m in re.findall(r"roaring\.NewFileBitmap\(([^)]*)\)", body)],
        "expected_counts": [int(k) for k in re.findall(r"n != (\d+)", body)],
        "optimized": re.findall(r"(bm\d)\.Optimize\(\)", body),
    }
body = func_body(open(os.path.join(REF, "bsi_test.go")).read(), "TestBSIAddCases")
for m in re.finditer(r"positions:\s*\[\]uint64\{([^}]*)\},\s*a:\s*\[\]uint64\{([^}]*)\},\s*b:\s*\[\]uint64\{([^}]*)\}", body):
    out["bsi_add_cases"].append({"positions": ints(m.group(1)), "a": ints(m.group(2)), "b": ints(m.group(3))})
# TestOpLogWriteUnmarshal (roaring/roaring_internal_test.go): the ops it writes and reads back, in order
body = func_body(open(os.path.join(REF, "roaring", "roaring_internal_test.go")).read(), "TestOpLogWriteUnmarshal")
body = body[: body.index("// test each one separately")]
TYPES = {"opTypeAdd": 0, "opTypeRemove": 1, "opTypeAddBatch": 2, "opTypeRemoveBatch": 3}
out["op_log_ops"] = []
for m in re.finditer(r"typ:\s*(opType\w+),\s*(value:\s*(\d+)|values:\s*\[\]uint64\{([^}]*)\})", body):
    typ = TYPES[m.group(1)]
    out["op_log_ops"].append({"type": typ, "value": int(m.group(3))} if m.group(3) is not None else {"type": typ, "values": ints(m.group(4))})
json.dump(out, open(os.path.join(HERE, "literal_vectors.json"), "w"), indent=1)
print({k: len(v["new_file_bitmap_literals"]) for k, v in out["intersection_count"].items()}, len(out["bsi_add_cases"]), "bsi add cases")
