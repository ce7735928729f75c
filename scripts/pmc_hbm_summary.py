"""Per-kernel HBM bytes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py, as
profiles/rNN_pmc_hbm_bytes.txt.   python scripts/pmc_hbm_summary.py <dir with pmc_fetch/ and pmc_write/> [title]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else root


def per_kernel(sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[(r["Kernel_Name"], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
print(f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `{title}`, one MI355X.")
print("FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the fetched bytes (MI355X_MICROARCH.md, HBM section), so")
print("read = 2 x FETCH_SIZE KiB.  Averages per launch over the launches of the pass, grouped by workgroup count.")
print()
print("%-75s %7s %9s %14s %14s" % ("kernel", "blocks", "launches", "read_bytes", "write_bytes"))
for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
    f = fetch[key]
    w = write.get(key, [0.0])
    name, blocks = key
    print("%-75s %7d %9d %14.0f %14.0f" % (name[:75], blocks, len(f), 2 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)))
