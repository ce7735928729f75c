#!/bin/bash
# one gpurun call of round 6: scripts/r6_call.sh <tag> <what> [...]   (output under gpurun_out/<tag>/)
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
for what in "$@"; do
case $what in
ring_smoke)   # first contact: small parity run of the ring kernel, then config 3's row pairs
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ring" > $O/parity_ring.log 2>&1; echo "parity_ring rc $?"; tail -3 $O/parity_ring.log ;;
ring_pairs)
  timeout 300 python scripts/bench_pairs.py --shards ${SHARDS:-256} --iters 20 --only-count --variants "${VARS:-pair_kernels=2;pair_kernels=3;pair_kernels=3,ring_geom=1;pair_kernels=3,ring_geom=2;pair_kernels=3,ring_geom=0,ring_nt=1}" --out $O/pairs_${SHARDS:-256}.json > $O/pairs_${SHARDS:-256}.log 2>&1; echo "pairs rc $?"; grep -h '"us"\|Error\|error\|assert' $O/pairs_${SHARDS:-256}.log | head -20 ;;
ring_fuzz)
  timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fuzz_struct.py -x -q -m gpu -k "ring or 3" > $O/fuzz_ring.log 2>&1; echo "fuzz_ring rc $?"; tail -3 $O/fuzz_ring.log ;;
ring_pmc)   # SQ counters of the ring kernel (three rocprofv3 --pmc passes)
  bash scripts/fused_pmc.sh $TAG/pmc_ring ${PMC_SHARDS:-256} "${PMC_OPTS:-pair_kernels=3}" pairs_pmc.py icount3 > $O/pmc_ring.txt 2>&1; cat $O/pmc_ring.txt | head -40 ;;
collective)  # host cost of one collective per headline step, on one GPU
  timeout 600 python scripts/collective_host_cost.py --out $O/collective_host_cost.json > $O/collective.log 2>&1; echo "collective rc $?"; tail -12 $O/collective.log ;;
fuzz)  # the seeded GPU parity tests re-rolled with fresh seeds
  bash scripts/fuzz_parity.sh $O ${SEEDS:-0x5eed6001 0x5eed6002 0x5eed6003 0x5eed6004} > $O/fuzz.out 2>&1; grep -h "== seed\|passed\|failed\|error" $O/fuzz_parity.log | head -20 ;;
gpu_tests)
  timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu_tests rc $?"; tail -3 $O/gpu_tests.log ;;
bench)
  timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json; cp bench_detail.json $O/bench_detail.json 2>/dev/null ;;
pair_tests)  # every test file that drives the pair kernels
  timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_struct.py tests/test_gpu_fullsize.py tests/test_gpu_queries.py -x -q -m gpu -n ${WORKERS:-4} > $O/pair_tests.log 2>&1; echo "pair_tests rc $?"; tail -3 $O/pair_tests.log ;;
ab)  # the pair kernels of several library builds side by side (LIBS, SHARDS, OPS as scripts/ab_pairs.sh)
  TAG=$TAG bash scripts/ab_pairs.sh ;;
first_launches)  # slow first launches after an upload / after idling
  timeout 900 python scripts/first_launches.py --shards ${SHARDS:-512} --out $O/first_launches.json > $O/first_launches.txt 2>&1; echo "first_launches rc $?"; grep -v "^\[" $O/first_launches.txt | tail -20 ;;
misc)  # rocprofv3 per-grid trace of the kernels the bench line does not reach
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/misc_trace -- python $R/scripts/profile_misc.py > $O/misc.log 2>&1; cd $R
  f=$(find $O/misc_trace -name "*kernel_trace.csv" | head -1); python scripts/kernel_trace_by_grid.py $f 3 > $O/misc_kernel_trace_by_grid.csv; grep -h "k_bsi_values\|k_bsi_cell\|k_bsi_add\|outlier" $O/misc_kernel_trace_by_grid.csv | head ;;
distinct_tests)
  timeout 600 python -m pytest tests/test_gpu_queries.py -x -q -m gpu -k "distinct or bsi" > $O/distinct_tests.log 2>&1; echo "distinct_tests rc $?"; tail -3 $O/distinct_tests.log ;;
*) echo "unknown: $what" ;;
esac
done
