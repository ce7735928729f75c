#!/bin/bash
# round 6's evidence in one gpurun call (the GPU suite and the fuzz seeds ran in their own calls): bench.py at N = 1 and N = 2 (gloo, one
# GPU), the same command under rocprofv3 (kernel trace -> per-grid summary with outlier flags), the PMC FETCH / WRITE passes,
# profile_misc.py per grid, the pair kernels' SQ counters.  Outputs under gpurun_out/r6p/, the summaries go to profiles/r06_*.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cp bench_detail.json $O/bench_detail.json
timeout 400 python bench.py --gpus 2 --steps 100 > $O/bench_n2.json 2> $O/bench_n2.err
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 200 --repeats 5 --no-cpu-baseline > $O/bench_prof.json 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python $R/bench.py --steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --cold-sets 1 --shards4 128 --shards4-mixed 128 --shards4-total 0 --shards4-mixed-total 0 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python $R/bench.py --steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --cold-sets 1 --shards4 128 --shards4-mixed 128 --shards4-total 0 --shards4-mixed-total 0 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/misc -o misc -- python $R/scripts/profile_misc.py 64 > $O/misc.json 2> /dev/null
cd $R
f=$(find $O/kt -name "*kernel_trace.csv" | head -1); python scripts/kernel_trace_by_grid.py $f 3 > $O/kernel_trace_by_grid.csv
f=$(find $O/misc -name "*kernel_trace.csv" | head -1); python scripts/kernel_trace_by_grid.py $f 3 > $O/misc_kernel_trace_by_grid.csv
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_trace_stats.csv 2>/dev/null
python scripts/pmc_hbm_summary.py $O "bench.py --steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --cold-sets 1 --shards4 128 --shards4-mixed 128 --shards4-total 0 --shards4-mixed-total 0 (round 6)" > $O/pmc_hbm_bytes.txt 2>&1
bash scripts/fused_pmc.sh r6p/pmc_pairs 256 "pair_kernels=2" pairs_pmc.py icount2 > $O/pmc_pairs.txt 2>&1
timeout 120 python scripts/bsi_bench.py 2>&1 | grep -v amdgpu.ids > $O/bsi_bench.txt
grep -c OUTLIER $O/kernel_trace_by_grid.csv; grep "^# outlier" $O/kernel_trace_by_grid.csv | head -5
head -c 700 $O/bench_n1.json; echo; head -c 400 $O/bench_n2.json; echo
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/misc $O/pmc_pairs
ls $O
