#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs (one counter pass per directory) per kernel:
average counter value per dispatch, and — for FETCH_SIZE / WRITE_SIZE, reported in KiB — the
bytes per dispatch with the gfx950 FETCH_SIZE half-count correction of MI355X_MICROARCH.md
(§HBM: FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced stream: double it).

    python scripts/pmc_summary.py gpurun_out/pmc_fetch/r_counter_collection.csv gpurun_out/pmc_write/r_counter_collection.csv
"""
import collections
import csv
import sys


def main():
    agg = collections.defaultdict(list)
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print(f"{'kernel':70s} {'counter':12s} {'n':>6s} {'avg':>16s} {'bytes/dispatch':>18s}")
    for (k, c), v in sorted(agg.items()):
        avg = sum(v) / len(v)
        b = ""
        if c == "FETCH_SIZE":
            b = f"{2 * avg * 1024:18.0f}"
        elif c == "WRITE_SIZE":
            b = f"{avg * 1024:18.0f}"
        print(f"{k:70s} {c:12s} {len(v):6d} {avg:16.3f} {b}")


if __name__ == "__main__":
    main()
