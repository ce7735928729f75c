"""Per-kernel launch statistics of a `rocprofv3 --kernel-trace` CSV, grouped by GRID SIZE (the stock --stats summary
averages all launches of a kernel name together, so that a half-size launch by another section of the same command
pulls the average of the full-size one down — VERDICT r3, evidence hygiene).

    python scripts/kernel_trace_by_grid.py <..._kernel_trace.csv> [min_calls=1] > profiles/rNN_kernel_trace_by_grid.csv

Round 6: the last two columns are max / avg and a flag for ratios > 5 (one 36 ms launch of a 1 ms kernel doubled a per-grid
AVERAGE in round 5's file and nobody saw it); every flagged launch is then listed on stderr-free '#' lines below the table with
its start time, the gap to the launch before it on the device and that launch's name: what an outlier followed says what it was.
"""
import collections
import csv
import re
import sys

path = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
acc = collections.defaultdict(list)
launches = []  # (start ns, end ns, key)
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"])  # drop the argument list
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    key = (name, grid // max(wg, 1), wg)
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    launches.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), key))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups", "workgroup_size", "calls", "total_us", "avg_us", "median_us", "min_us", "max_us", "max_over_avg", "flag"])
rows = []
for (name, blocks, wg), ts in acc.items():
    if len(ts) < min_calls:
        continue
    ts.sort()
    rows.append((sum(ts), name, blocks, wg, len(ts), ts))
for tot, name, blocks, wg, n, ts in sorted(rows, reverse=True):
    ratio = ts[-1] / (tot / n) if tot else 0.0
    w.writerow([name, blocks, wg, n, f"{tot:.1f}", f"{tot / n:.2f}", f"{ts[n // 2]:.2f}", f"{ts[0]:.2f}", f"{ts[-1]:.2f}", f"{ratio:.2f}", "OUTLIER" if ratio > 5 else ""])
# the outliers themselves: when, after what
launches.sort()
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
t0 = launches[0][0] if launches else 0
for i, (st, en, key) in enumerate(launches):
    d = (en - st) / 1e3
    if len(acc[key]) >= max(min_calls, 4) and d > 5 * med[key] and d > 50:
        prev = launches[i - 1] if i else None
        gap = (st - prev[1]) / 1e3 if prev else 0.0
        print(f"# outlier: {key[0]} x{key[1]} took {d:.1f} us (median {med[key]:.1f}) at t = {(st - t0) / 1e6:.3f} ms, launch #{i} of {len(launches)}; "
              f"the launch before it: {prev[2][0] if prev else '-'} x{prev[2][1] if prev else 0}, ended {gap:.1f} us earlier")
