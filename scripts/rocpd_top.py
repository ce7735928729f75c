#!/usr/bin/env python3
"""Print the per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db).

    python scripts/rocpd_top.py gpurun_out/prof/r_results.db [> profiles/rNN_kernel_trace_stats.txt]
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
    rows = list(cur.execute("select * from top_kernels"))
    print("# columns:", ", ".join(cols))
    name_i = cols.index("name")
    print(f"{'kernel':110s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows:
        d = dict(zip(cols, r))
        # the top_kernels view reports durations in microseconds
        tot = d.get("total_duration", 0)
        avg = d.get("average", 0)
        print(f"{str(r[name_i])[:110]:110s} {d.get('total_calls', 0):7d} {tot:12.3f} {avg:10.3f} {d.get('percentage', 0):6.2f}")


if __name__ == "__main__":
    main()
