#!/usr/bin/env python3
"""Two outputs of scripts/bench_ctops.py (two library builds, one box) side by side: every cell whose time moved by more than 5 %.

    python scripts/ctops_ab.py old.json new.json [op ...]
"""
import json
import sys

a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
ops = sys.argv[3:] or ["intersectionCount", "intersect", "union", "difference", "xor"]


def cells(d):
    out = {}
    for name, c in d["cells"].items():  # "A/B" -> {op: {"us": ...}}
        x, y = name.split("/")
        for op in ops:
            if isinstance(c.get(op), dict) and "us" in c[op]:
                out[(x, y, op)] = c[op]["us"]
    return out


ca, cb = cells(a), cells(b)
common = sorted(set(ca) & set(cb))
slower = [(k, ca[k], cb[k]) for k in common if cb[k] > 1.05 * ca[k]]
faster = [(k, ca[k], cb[k]) for k in common if cb[k] < 0.95 * ca[k]]
print(f"{len(common)} cells compared ({', '.join(ops)}); > 5 % slower: {len(slower)}; > 5 % faster: {len(faster)}")
for tag, rows in (("SLOWER", slower), ("faster", faster)):
    for (x, y, op), u, v in sorted(rows, key=lambda r: r[2] / r[1], reverse=(tag == "SLOWER")):
        print(f"  {tag} {x:12s} x {y:12s} {op:18s} {u:9.1f} -> {v:9.1f} us ({v / u:.3f} x)")
if common:
    import statistics

    print(f"geometric mean of new / old: {statistics.geometric_mean([cb[k] / ca[k] for k in common]):.4f}")
