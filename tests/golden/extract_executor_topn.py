#!/usr/bin/env python3
"""Extract the executor-level TopN known answers of executor_test.go mechanically: TestExecutor_Execute_TopN
(the RowIDColumnID sub-test; the keyed variants repeat it through the key translator, which is off the path),
TestExecutor_Execute_TopN_fill, _TopN_fill_small and _TopN_Src (:1846-2200).  Each case = the bits set (row, column;
`ShardWidth` expressions evaluated with ShardWidth = 2^20), the bits of the source row of `TopN(f, Row(other=..), n=..)`
when there is one, n, and the expected pairs.  These are the vectors that pin executeTopN's two passes
(executor.go:2779-2864): _fill_small is the cross-shard candidate case (five shards, n = 1 -> {0: 5}).

    python tests/golden/extract_executor_topn.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SHARD_WIDTH = 1 << 20


def col_value(expr: str) -> int:
    """`ShardWidth+2`, `(5*ShardWidth)+100`, `2*ShardWidth+1`, `0`"""
    e = expr.strip()
    assert re.fullmatch(r"[0-9ShardWidth+*() ]+", e), e
    return int(eval(e.replace("ShardWidth", str(SHARD_WIDTH)), {"__builtins__": {}}))


def func_body(src, name):
    m = re.search(r"^func %s\(t \*testing\.T\) \{" % name, src, re.M)
    end = src.index("\n}\n", m.end())
    return src[m.end(): end], src.count("\n", 0, m.start()) + 1


def expected_pairs(text):
    de = re.search(r"Pairs: \[\]pilosa\.Pair\{(.*?)\},\s*Field:", text, re.S)
    return [[int(a), int(b)] for a, b in re.findall(r"\{ID: (\d+), Count: (\d+)\}", de.group(1))]


def main():
    src = open(os.path.join(REF, "executor_test.go")).read()
    cases = []
    # ---- TestExecutor_Execute_TopN / RowIDColumnID: bits come from a PQL Set(...) script built by string concatenation
    body, line = func_body(src, "TestExecutor_Execute_TopN")
    sub = body[body.index('t.Run("RowIDColumnID"'): body.index('t.Run("RowIDColumnKey"')]
    script = sub[sub.index("Query: `") + 8: sub.index("`}); err != nil")]
    script = re.sub(r"` \+ strconv\.Itoa\(([^`]*?)\) \+ `", lambda m: str(col_value(m.group(1))), script)
    bits = {"f": [], "other": []}
    for col, fld, row in re.findall(r"Set\((\d+), (\w+)=(\d+)\)", script):
        bits[fld].append([int(row), int(col)])
    q = re.search(r"Query: `TopN\(f, n=(\d+)\)`", sub)
    cases.append({"test": "TestExecutor_Execute_TopN/RowIDColumnID", "line": line, "bits": bits["f"], "src_bits": None, "n": int(q.group(1)),
                  "expected": expected_pairs(sub[q.end():])})
    # ---- the SetBit forms
    for name in ("TestExecutor_Execute_TopN_fill", "TestExecutor_Execute_TopN_fill_small", "TestExecutor_Execute_TopN_Src"):
        body, line = func_body(src, name)
        f_bits, o_bits = [], []
        for fld, row, col in re.findall(r'hldr\.SetBit\(c\.Idx\(\), "(\w+)", (\d+), ([^)\n]*)\)', body):
            (f_bits if fld == "f" else o_bits).append([int(row), col_value(col)])
        q = re.search(r"Query: `TopN\(f, (?:Row\(other=(\d+)\), )?n=(\d+)\)`", body)
        src_bits = None
        if q.group(1) is not None:
            src_bits = [c for r, c in o_bits if r == int(q.group(1))]
        cases.append({"test": name, "line": line, "bits": f_bits, "src_bits": src_bits, "n": int(q.group(2)), "expected": expected_pairs(body[q.end():])})
    with open(os.path.join(OUT, "executor_topn_vectors.json"), "w") as f:
        json.dump({"source": "executor_test.go:1846-2200", "shard_width": SHARD_WIDTH, "cases": cases}, f, indent=1)
    for c in cases:
        print(c["test"], "n =", c["n"], "bits", len(c["bits"]), "src", c["src_bits"], "->", c["expected"])


if __name__ == "__main__":
    main()
