#!/bin/bash
# per-kernel durations of a driver script under rocprofv3 --kernel-trace --stats
#   scripts/kernel_times.sh <out dir under gpurun_out> <driver.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
D=$1; shift
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python $R/scripts/$D "$@" > $O/kt.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f} us")
PY
