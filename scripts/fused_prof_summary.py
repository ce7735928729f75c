"""Summarise the cycle stamps printed by scripts/fused_prof.py (stderr of the instrumented kernel)."""
import collections
import re
import sys

rows = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    m = re.match(r"st\s+(\d+) ([cp])(\d+)\s+(.*)", l)
    if m:
        rows[int(m.group(1))][int(m.group(3))] = [int(x) for x in m.group(4).split()]
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for st in sorted(rows)[:lim]:
    r = rows[st]
    t0 = min(v[0] for v in r.values() if v[0] >= 0)
    t1 = max(v[5] for v in r.values())
    cons = [r[w] for w in range(4)]
    prod = [r[w] for w in range(4, 16)]
    print(f"stage {st:2d}: len {t1 - t0:6d} | consumers done {[c[1] - t0 for c in cons]} | producers: settle+bitmaps {[p[1] - p[0] for p in prod]} "
          f"loads {[p[2] - p[1] for p in prod]} arrays {[p[3] - p[2] for p in prod]} runs+lists {[p[4] - p[3] for p in prod]} arrive {[p[4] - t0 for p in prod]}")
