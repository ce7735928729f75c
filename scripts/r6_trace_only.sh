#!/bin/bash
# bench.py at N = 1 and the same command under rocprofv3 --kernel-trace --stats (per-grid summary with outlier flags): the two records that must
# agree on the headline kernel's launch duration.  Outputs under gpurun_out/r6t/, copied to profiles/r06_*.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6t
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cp bench_detail.json $O/bench_detail.json
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 200 --repeats 5 --no-cpu-baseline > $O/bench_prof.json 2> /dev/null
cd $R
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); python scripts/kernel_trace_by_grid.py $f 3 > $O/kernel_trace_by_grid.csv
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_trace_stats.csv 2>/dev/null
rm -rf $O/kt
grep "icount_dense" $O/kernel_trace_by_grid.csv; head -c 600 $O/bench_n1.json; echo
