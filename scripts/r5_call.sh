# one gpurun call of round 5: STEPS is a space-separated list of step names (default: all)
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r5a}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
STEPS=${STEPS:-"pytest pairs ctops"}
for s in $STEPS; do
case $s in
pytest) (timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} 2>&1 | tail -25) > $O/pytest.log ;;
pytest_sel) (timeout ${PYTEST_TIMEOUT:-900} python -m pytest ${PYTEST_SEL} -m gpu -q 2>&1 | tail -60) > $O/pytest_sel.log ;;
pytest_full) (timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=10 2>&1 | tail -40) > $O/pytest_full.log ;;
pairs) timeout 300 python scripts/bench_pairs.py --out $O/pairs.json ${PAIRS_ARGS:-} > $O/pairs.txt 2> $O/pairs.err ;;
ctops_small) timeout 400 python scripts/bench_ctops.py --rows 1024 --iters 20 --out $O/ctops_small.json --only Empty,Ary1,Ary16,Ary256,Ary512,BM4096,RunFull,Run16,Run256 2>&1 | grep -v amdgpu.ids | tail -40 > $O/ctops_small.txt ;;
ctops) timeout 600 python scripts/bench_ctops.py --rows 1024 --iters 20 --out $O/ctops.json 2>&1 | grep -v amdgpu.ids | tail -80 > $O/ctops.txt ;;
bench) timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ;;
bench2) timeout 400 python bench.py --gpus 2 --steps 100 > $O/bench_n2.json 2> $O/bench_n2.err ;;
misc) (cd /tmp; export TMPDIR=/tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/misc -o misc -- python $R/scripts/profile_misc.py 64 > $O/misc.json 2> /dev/null; python3 $R/scripts/kernel_trace_by_grid.py $O/misc/misc_kernel_trace.csv 1 > $O/misc_kernel_trace_by_grid.csv) ;;
benchprof) (cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 200 --repeats 5 --no-cpu-baseline --shards4-total 1024 --shards4-mixed-total 1024 > $O/bench_prof.json 2> /dev/null; python3 $R/scripts/kernel_trace_by_grid.py $O/kt/bench_kernel_trace.csv 3 > $O/kernel_trace_by_grid.csv) ;;
pmc_pairs) KF=${KF:-icount}; bash scripts/fused_pmc.sh $TAG/pmc_v1 64 pair_kernels=1 pairs_pmc.py $KF > $O/pmc_pairs_v1.txt 2>&1; bash scripts/fused_pmc.sh $TAG/pmc_v2 64 pair_kernels=2 pairs_pmc.py $KF > $O/pmc_pairs_v2.txt 2>&1 ;;
bsi) (timeout 200 python scripts/bsi_bench.py 2>&1 | grep -v amdgpu.ids > $O/bsi_bench.txt) ;;
pmc_hbm) (cd /tmp; export TMPDIR=/tmp; BA="--steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --cold-sets 1 --shards4 128 --shards4-mixed 128 --shards4-total 0 --shards4-mixed-total 0"; timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python $R/bench.py $BA > /dev/null 2>&1; timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python $R/bench.py $BA > /dev/null 2>&1; python3 $R/scripts/pmc_hbm_summary.py $O "bench.py $BA (round 5)" > $O/pmc_hbm_bytes.txt 2>&1) ;;
pmc_pairs4) bash scripts/fused_pmc.sh $TAG/pmc_pairs ${PMC_SHARDS:-256} pair_kernels=2 pairs_pmc.py icount2 > $O/pmc_pairs_shipped.txt 2>&1 ;;
pmc_scatter) bash scripts/fused_pmc.sh $TAG/pmc_scatter 256 0 scatter_pmc.py ${KF:-k_} > $O/pmc_scatter.txt 2>&1 ;;
spbsweep) timeout 300 python scripts/matrix_spb_sweep.py ${SWEEP_ARGS:-1024 6} 2> $O/spb_sweep.err | grep -v amdgpu.ids > $O/spb_sweep.json ;;
small) timeout 300 python scripts/small_shapes_ab.py ${SMALL_ARGS:-64 7} 2> $O/small_shapes.err | grep -v amdgpu.ids > $O/small_shapes.json ;;
scatter_ab) (for i in 1 2; do for lib in "" build_variants/${AB_VARIANT:-r4_container_dealing}/libfbk.so; do FBK_LIB_PATH=${lib:+$R/$lib} timeout 200 python scripts/scatter_ab.py 2>> $O/scatter_ab.err | grep "^{" >> $O/scatter_ab.jsonl; done; done) ;;
pmc_fused) bash scripts/fused_pmc.sh $TAG/pmc_fused 256 0 fused_pmc.py count_matrix_fused > $O/pmc_fused_shipped.txt 2>&1 ;;
fuzz) bash scripts/fuzz_parity.sh $O ${FUZZ_SEEDS:-0x5eed4001 0x5eed4002 0x5eed4003} > /dev/null 2>&1 ;;
fusedprof) (timeout 200 python scripts/fused_prof.py 256 0 2 ${PROF_CFG:-4} 2>&1 | grep -v amdgpu.ids) > $O/fused_prof.txt ;;
pairs_spw) timeout 300 python scripts/bench_pairs.py --shards ${SPW_SHARDS:-256} --iters 20 --only-count --variants 'pair_kernels=2,pair_spw=1;pair_kernels=2,pair_spw=2;pair_kernels=2,pair_spw=4;pair_kernels=2,pair_spw=1' --out $O/pairs_spw_${SPW_SHARDS:-256}.json > $O/pairs_spw.log 2>&1 ;;
setop_ablate) V=''; for a in ${ABL_LIST:-0 8 16 32 56 64 128 192 256 0}; do V="$V;pair_kernels=2,pair_ablate=$a"; done; timeout 400 python scripts/bench_pairs.py --shards 256 --iters 20 --ops "${ABL_OPS:-intersectionCount,intersect + optimize(),intersect}" --variants "${V#;}" --out $O/setop_ablate.json > $O/setop_ablate.log 2>&1 ;;
fused_ab) timeout 500 python scripts/fused_ab.py ${FUSED_AB_ARGS:-256 1024} > $O/fused_ab.json 2> $O/fused_ab.err ;;
fused_ablate) (for c in ${ABL_CFGS:-4 3}; do timeout 300 python scripts/fused_ablate.py $c 2>> $O/fused_ablate.err | grep "^{" >> $O/fused_ablate.jsonl; done) ;;
*) echo "unknown step $s" ;;
esac
done
ls -R $O | head -40
for f in $O/pytest.log $O/pytest_sel.log $O/pytest_full.log $O/bench_n1.err $O/bench_n2.err; do [ -f $f ] && tail -60 $f; done
