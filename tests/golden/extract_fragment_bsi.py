#!/usr/bin/env python3
"""Extract the reference's BSI Range / Sum known-answer cases into a JSON fixture.

Source: fragment_internal_test.go TestFragment_Sum (:452-521), TestFragment_Range
(:606-916), TestFragment_MinMax (:524-604) and TestIntLTRegression (:3738-3756).  Each subtest is a list of
setValue(col, bitDepth, value) calls followed by queries with literal expected columns;
only those literals are recorded (with their line), never code.

    python tests/golden/extract_fragment_bsi.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def num(tok, consts):
    tok = tok.strip()
    if tok in consts:
        return consts[tok]
    return int(tok, 0)


def main():
    src = open(os.path.join(REF, "fragment_internal_test.go")).read()
    lines = src.split("\n")

    def line_of(idx):
        return src.count("\n", 0, idx) + 1

    cases = []
    # ---- TestFragment_Range: split into t.Run blocks
    m0 = re.search(r"^func TestFragment_Range\(", src, re.M)
    m1 = re.search(r"^func ", src[m0.end():], re.M)
    body = src[m0.start(): m0.end() + m1.start()]
    base = m0.start()
    consts = {"bitDepth": int(re.search(r"const bitDepth = (\d+)", body).group(1))}
    runs = list(re.finditer(r't\.Run\("(\w+)", func', body))
    for i, r in enumerate(runs):
        blk = body[r.start(): runs[i + 1].start() if i + 1 < len(runs) else len(body)]
        off = base + r.start()
        vals = [[int(a, 0), num(b, consts), int(c, 0)] for a, b, c in re.findall(r"f\.setValue\(tx, (\w+), (\w+), (-?\w+)\)", blk)]
        queries = []
        for q in re.finditer(
            r"f\.(rangeOp|rangeBetween|rangeBetweenUnsigned)\(tx, ([^)]*(?:\([^)]*\))?[^)]*)\); err != nil \{.*?\[\]uint64\{([^}]*)\}",
            blk,
            re.S,
        ):
            kind, args, exp = q.group(1), q.group(2), q.group(3)
            expl = [int(x, 0) for x in exp.replace(" ", "").split(",") if x]
            ln = line_of(off + q.start())
            if kind == "rangeOp":
                op, depth, pred = [a.strip() for a in args.split(",")]
                queries.append({"kind": "rangeOp", "op": op.replace("pql.", ""), "depth": num(depth, consts), "pred": int(pred, 0), "exp": expl, "line": ln})
            elif kind == "rangeBetween":
                depth, lo, hi = [a.strip() for a in args.split(",")]
                queries.append({"kind": "rangeBetween", "depth": num(depth, consts), "lo": int(lo, 0), "hi": int(hi, 0), "exp": expl, "line": ln})
            else:
                mm = re.match(r"NewRow\(([^)]*)\), (\w+), (\w+), (\w+)", args.strip())
                queries.append(
                    {
                        "kind": "rangeBetweenUnsigned",
                        "filter": [int(x, 0) for x in mm.group(1).replace(" ", "").split(",") if x],
                        "depth": num(mm.group(2), consts),
                        "lo": int(mm.group(3), 0),
                        "hi": int(mm.group(4), 0),
                        "exp": expl,
                        "line": ln,
                    }
                )
        for q in re.finditer(
            r"f\.(rangeLTUnsigned|rangeGTUnsigned)\(tx, NewRow\(([^)]*)\), (\w+), (\w+), (true|false)\); err != nil \{.*?\[\]uint64\{([^}]*)\}",
            blk,
            re.S,
        ):
            queries.append(
                {
                    "kind": q.group(1),
                    "filter": [int(x, 0) for x in q.group(2).replace(" ", "").split(",") if x],
                    "depth": num(q.group(3), consts),
                    "pred": int(q.group(4), 0),
                    "allow_eq": q.group(5) == "true",
                    "exp": [int(x, 0) for x in q.group(6).replace(" ", "").split(",") if x],
                    "line": line_of(off + q.start()),
                }
            )
        cases.append({"test": "TestFragment_Range/" + r.group(1), "line": line_of(off), "values": vals, "queries": queries})
    # ---- TestIntLTRegression (:3738): setValue(1, 6, 33); rangeOp(LT, 6, 33) must be empty
    m = re.search(r"^func TestIntLTRegression\(", src, re.M)
    blk = src[m.start(): m.start() + 800]
    v = re.search(r"f\.setValue\(tx, (\d+), (\d+), (\d+)\)", blk)
    q = re.search(r"f\.rangeOp\(tx, pql\.(\w+), (\d+), (\d+)\)", blk)
    cases.append(
        {
            "test": "TestIntLTRegression",
            "line": line_of(m.start()),
            "values": [[int(v.group(1)), int(v.group(2)), int(v.group(3))]],
            "queries": [{"kind": "rangeOp", "op": q.group(1), "depth": int(q.group(2)), "pred": int(q.group(3)), "exp": [], "line": line_of(m.start() + q.start())}],
        }
    )
    # ---- TestFragment_Sum (:452): values + (filter -> sum, count)
    m = re.search(r"^func TestFragment_Sum\(", src, re.M)
    blk = src[m.start(): m.start() + 2600]
    depth = int(re.search(r"const bitDepth = (\d+)", blk).group(1))
    vals = [[int(a), depth, int(b)] for a, b in re.findall(r"\{(\d+), (-?\d+)\},", blk)]
    sums = [
        {"filter": None, "count": 5, "sum": 382 + 300 - 600 + 2818 + 300, "line": line_of(m.start() + blk.index('"NoFilter"'))},
        {"filter": [2000, 4000, 5000], "count": 2, "sum": 300 + 300, "line": line_of(m.start() + blk.index('"WithFilter"'))},
    ]
    # sanity: the literals above must literally be in the source
    assert "int64(382+300-600+2818+300)" in blk and "NewRow(2000, 4000, 5000)" in blk and "int64(300+300)" in blk
    assert "n != 5" in blk and "n != 2" in blk
    sum_case = {"test": "TestFragment_Sum", "line": line_of(m.start()), "values": vals, "sums": sums}

    # ---- TestFragment_MinMax (:524-604): values + per-filter (min|max, count) tables
    m = re.search(r"^func TestFragment_MinMax\(", src, re.M)
    m1 = re.search(r"^func ", src[m.end():], re.M)
    blk = src[m.start(): m.end() + m1.start()]
    depth = int(re.search(r"const bitDepth = (\d+)", blk).group(1))
    vals = [[int(a), depth, int(b)] for a, b in re.findall(r"f\.setValue\(tx, (\d+), bitDepth, (-?\d+)\)", blk)]
    minmax = {"test": "TestFragment_MinMax", "line": line_of(m.start()), "values": vals, "depth": depth}
    for kind in ("Min", "Max"):
        k0 = blk.index('t.Run("%s"' % kind)
        sub = blk[k0: blk.index("for i, test := range tests", k0)]
        rows = []
        for q in re.finditer(r"\{filter: (nil|NewRow\(([^)]*)\)), exp: (-?\d+), cnt: (\d+)\}", sub):
            flt = None if q.group(1) == "nil" else [int(x, 0) for x in q.group(2).replace(" ", "").split(",") if x]
            rows.append({"filter": flt, "exp": int(q.group(3)), "cnt": int(q.group(4)), "line": line_of(m.start() + k0 + q.start())})
        minmax[kind.lower()] = rows
    assert len(minmax["min"]) == 6 and len(minmax["max"]) == 6 and len(vals) == 7

    with open(os.path.join(OUT, "fragment_bsi_cases.json"), "w") as f:
        json.dump(
            {"source": "fragment_internal_test.go:452-604, 606-916, 3738-3756", "range_cases": cases, "sum_case": sum_case, "minmax_case": minmax},
            f,
            indent=1,
        )
    print("minmax:", minmax)
    nq = sum(len(c["queries"]) for c in cases)
    print(f"fragment_bsi_cases.json: {len(cases)} range subtests, {nq} queries; sum case with {len(vals)} values")
    for c in cases:
        print("  ", c["test"], len(c["values"]), "values", [(q.get("op", q["kind"]), q.get("pred", (q.get("lo"), q.get("hi"))), q["exp"]) for q in c["queries"]])
    del lines


if __name__ == "__main__":
    main()
