#!/usr/bin/env python3
"""Extract the executor-level set-op vectors of executor_test.go (:1236-1373): SetBit lists,
the PQL call and the expected columns / count.  Columns straddle ShardWidth, so these pin the
per-shard map + concatenation (executor.go:1758-1769).

    python tests/golden/extract_executor_vectors.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SW = 1 << 20


def ev(expr: str) -> int:
    return int(eval(expr.replace("ShardWidth", str(SW)), {"__builtins__": {}}))


def main():
    src = open(os.path.join(REF, "executor_test.go")).read()
    cases = []
    for name in ("Difference", "Intersect", "Union", "Xor", "Count"):
        m = re.search(r"^func TestExecutor_Execute_%s\(t \*testing\.T\) \{" % name, src, re.M)
        end = src.index("\n}\n", m.end())
        body = src[m.end(): end]
        sub = body[: body.index("\n\t})")] if "\n\t})" in body else body  # first subtest (RowIDColumnID)
        bits = [(f, int(r), ev(c)) for f, r, c in re.findall(r'hldr\.SetBit\(c\.Idx\(\), "(\w+)", (\d+), ([^)]+)\)', sub)]
        q = re.search(r"Query: `([^`]+)`", sub).group(1)
        exp_cols = re.search(r"\[\]uint64\{([^}]*)\}", sub)
        exp_n = re.search(r"res\.Results\[0\] != uint64\((\d+)\)", sub)
        case = {"test": f"TestExecutor_Execute_{name}", "line": src.count("\n", 0, m.start()) + 1, "bits": bits, "query": q}
        if exp_n:
            case["count"] = int(exp_n.group(1))
        else:
            case["columns"] = [ev(x) for x in exp_cols.group(1).split(",") if x.strip()]
        cases.append(case)
    with open(os.path.join(OUT, "executor_vectors.json"), "w") as f:
        json.dump({"source": "executor_test.go:1236-1373", "shard_width": SW, "cases": cases}, f, indent=1)
    for c in cases:
        print(c["test"], c["query"], len(c["bits"]), "bits ->", c.get("columns", c.get("count")))


if __name__ == "__main__":
    main()
