#!/usr/bin/env python3
"""Extract the fragment.top known answers of fragment_internal_test.go mechanically:
TestFragment_Top, _TopN_Intersect, _TopN_Intersect_Large, _TopN_IDs, _Tanimoto and _Zero_Tanimoto
(:1148-1273, :1490-1538): the bits set (mustSetBits literals, or the one generator loop of the
_Large test, recognised by its shape), the source row, the topOptions and the expected pairs.

    python tests/golden/extract_topn_vectors.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
TESTS = ["TestFragment_Top", "TestFragment_TopN_Intersect", "TestFragment_TopN_Intersect_Large", "TestFragment_TopN_IDs", "TestFragment_Tanimoto",
         "TestFragment_Zero_Tanimoto"]


def ints(s):
    return [int(x) for x in re.findall(r"\d+", s)]


def main():
    src = open(os.path.join(REF, "fragment_internal_test.go")).read()
    cases = []
    for name in TESTS:
        m = re.search(r"^func %s\(t \*testing\.T\) \{" % name, src, re.M)
        end = src.index("\n}\n", m.end())
        body = src[m.end(): end]
        case = {"test": name, "line": src.count("\n", 0, m.start()) + 1}
        s = re.search(r"src := NewRow\(([^)]*)\)", body, re.S)
        case["src"] = ints(s.group(1)) if s else None
        rows = {}
        for row, cols in re.findall(r"f\.mustSetBits\(tx, (\d+), ([^)]*)\)", body):
            rows.setdefault(int(row), []).extend(ints(cols))
        gen = re.search(r"for i := uint64\(0\); i < (\d+); i\+\+ \{\s*for j := uint64\(0\); j < i; j\+\+ \{\s*addToBitmap\(bm, i, j\)", body)
        if gen:  # rows 0..n-1, row i holds columns 0..i-1
            case["generator"] = {"kind": "row_i_has_columns_below_i", "n": int(gen.group(1))}
        else:
            case["rows"] = {str(k): v for k, v in rows.items()}
        o = re.search(r"f\.top\(tx, topOptions\{([^;]*?)\}\); err", body, re.S).group(1)
        opt = {"N": 0, "MinThreshold": 0, "TanimotoThreshold": 0, "RowIDs": None, "Src": False}
        for key in ("N", "MinThreshold", "TanimotoThreshold"):
            k = re.search(r"\b%s: (\d+)" % key, o)
            if k:
                opt[key] = int(k.group(1))
        rid = re.search(r"RowIDs: \[\]uint64\{([^}]*)\}", o)
        if rid:
            opt["RowIDs"] = ints(rid.group(1))
        opt["Src"] = bool(re.search(r"\bSrc: src", o))
        case["options"] = opt
        tail = body[body.index("f.top(tx"):]
        de = re.search(r"reflect\.DeepEqual\(pairs, \[\]Pair\{(.*?)\}\) \{", tail, re.S)
        if de:
            case["expected"] = [[int(a), int(b)] for a, b in re.findall(r"\{ID: (\d+), Count: (\d+)\}", de.group(1))]
        else:
            n = int(re.search(r"len\(pairs\) != (\d+)", tail).group(1))
            got = {int(i): [int(a), int(b)] for i, a, b in re.findall(r"pairs\[(\d+)\] != \(Pair\{ID: (\d+), Count: (\d+)\}\)", tail)}
            assert sorted(got) == list(range(n)), (name, got)
            case["expected"] = [got[i] for i in range(n)]
        cases.append(case)
    with open(os.path.join(OUT, "topn_vectors.json"), "w") as f:
        json.dump({"source": "fragment_internal_test.go:1148-1273, 1490-1538", "cases": cases}, f, indent=1)
    for c in cases:
        print(c["test"], c["options"], "->", c["expected"][:4], "..." if len(c["expected"]) > 4 else "")


if __name__ == "__main__":
    main()
