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
ctops_ab)  # the reference's container-archetype matrix on two library builds (LIB_OLD: a build_variants/ name), every cell oracle-checked
  FBK_LIB_PATH=$R/build_variants/${LIB_OLD:-r5probe}/libfbk.so timeout 900 python scripts/bench_ctops.py --rows 1024 --iters 20 --out $O/ctops_old.json > $O/ctops_old.log 2>&1; echo "ctops old rc $?"
  timeout 900 python scripts/bench_ctops.py --rows 1024 --iters 20 --out $O/ctops_new.json > $O/ctops_new.log 2>&1; echo "ctops new rc $?"
  python scripts/ctops_ab.py $O/ctops_old.json $O/ctops_new.json > $O/ctops_ab.txt 2>&1; head -30 $O/ctops_ab.txt ;;
fresh_under_rocprof)  # the first run of a new query on fresh memory: the library's HIP events against rocprofv3's dispatch timestamps
  (cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/fresh_trace -- python $R/scripts/first_launches.py --shards ${SHARDS:-1024} --only-fresh --fresh-sleep ${FRESH_SLEEP:-10} --out $O/fresh.json > $O/fresh.txt 2>&1)
  grep "^H " $O/fresh.txt; f=$(find $O/fresh_trace -name "*kernel_trace.csv" | head -1); python scripts/kernel_trace_by_grid.py $f 3 > $O/fresh_by_grid.csv; grep "fusedq\|outlier" $O/fresh_by_grid.csv | head
  python - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "fusedq" in r["Kernel_Name"] and d > 5000:
        print("--- context of a %.0f us launch of k_count_matrix_fusedq (rocprofv3 clock):" % d)
        for q in rows[max(0, i - 8): i + 2]:
            print("   %-60s start +%10.1f us  dur %10.1f us" % (q["Kernel_Name"][:60], (int(q["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3))
PY
  rm -rf $O/fresh_trace ;;
libab)  # k_count_matrix_fusedq of several library builds (LIBS: names under build_variants/, or "product"), alternating, rows generated once
  timeout 600 python scripts/fused_lib_ab.py gen ${S3:-256} ${S4:-1024} > $O/libab_gen.log 2>&1; echo "gen rc $?"
  for round in 1 2 3; do for L in ${LIBS:-product}; do
    if [ $L = product ]; then timeout 300 python scripts/fused_lib_ab.py run $LIBAB_OPTS >> $O/libab.jsonl 2>> $O/libab.err
    else FBK_LIB_PATH=$R/build_variants/$L/libfbk.so timeout 300 python scripts/fused_lib_ab.py run $LIBAB_OPTS >> $O/libab.jsonl 2>> $O/libab.err; fi
  done; done
  python scripts/fused_lib_ab.py rm; cut -c1-400 $O/libab.jsonl ;;
fprof)  # cycle stamps of one block of k_count_matrix_fusedq (the -DFBK_EXPERIMENTS build), config 3 and config 4
  timeout 400 python scripts/fused_prof.py 256 0 0 3 2> $O/fprof_c3.txt > /dev/null; python scripts/fused_prof_summary.py $O/fprof_c3.txt 24 > $O/fprof_c3_summary.txt; head -12 $O/fprof_c3_summary.txt | cut -c1-300
  timeout 600 python scripts/fused_prof.py 1024 0 0 4 2> $O/fprof_c4.txt > /dev/null; python scripts/fused_prof_summary.py $O/fprof_c4.txt 24 > $O/fprof_c4_summary.txt; head -12 $O/fprof_c4_summary.txt | cut -c1-300 ;;
*) echo "unknown: $what" ;;
esac
done
