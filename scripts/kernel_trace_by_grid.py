"""Per-kernel launch statistics of a `rocprofv3 --kernel-trace` CSV, grouped by GRID SIZE (the stock --stats summary
averages all launches of a kernel name together, so that a half-size launch by another section of the same command
pulls the average of the full-size one down — VERDICT r3, evidence hygiene).

    python scripts/kernel_trace_by_grid.py <..._kernel_trace.csv> [min_calls=1] > profiles/rNN_kernel_trace_by_grid.csv
"""
import collections
import csv
import re
import sys

path = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
acc = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])  # drop the argument list
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    acc[(name, grid // max(wg, 1), wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups", "workgroup_size", "calls", "total_us", "avg_us", "median_us", "min_us", "max_us"])
rows = []
for (name, blocks, wg), ts in acc.items():
    if len(ts) < min_calls:
        continue
    ts.sort()
    rows.append((sum(ts), name, blocks, wg, len(ts), ts))
for tot, name, blocks, wg, n, ts in sorted(rows, reverse=True):
    w.writerow([name, blocks, wg, n, f"{tot:.1f}", f"{tot / n:.2f}", f"{ts[n // 2]:.2f}", f"{ts[0]:.2f}", f"{ts[-1]:.2f}"])
