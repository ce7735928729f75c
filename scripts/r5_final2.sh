# the last gpurun call of round 5 (4.5 GPU-minutes left): timings of three library builds first (nothing else on the device), then
# the whole -m gpu suite on the build with both code-size changes (build_variants/emitloop) and, beside it, the pair-kernel test
# files on the build with the count kernel's change only (the product library)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5y
mkdir -p $O
cd $R
date +%s > $O/t0
python -c "from oracle import pyoracle; pyoracle.build()" > $O/oracle_build.log 2>&1
OPS='intersectionCount,intersect + optimize(),difference + optimize()'
i=0
for v in old base emitloop old; do
  i=$((i + 1))
  lib=$R/build_variants/$v/libfbk.so
  [ "$v" = base ] && lib=""
  FBK_LIB_PATH=$lib timeout 40 python scripts/bench_pairs.py --shards 256 --iters 20 --ops "$OPS" --variants "pair_kernels=2" --out $O/${i}_$v.json > $O/${i}_$v.log 2>&1
  echo "$i $v rc $? $(grep -h '"us"' $O/${i}_$v.json | tr -d ' \n')" | tee -a $O/timings.txt
done
for v in old base; do
  i=$((i + 1))
  lib=$R/build_variants/$v/libfbk.so
  [ "$v" = base ] && lib=""
  FBK_LIB_PATH=$lib timeout 30 python scripts/bench_pairs.py --shards 64 --iters 40 --only-count --variants "pair_kernels=2" --out $O/${i}_${v}_2048.json > $O/${i}_${v}_2048.log 2>&1
  echo "$i $v 2048 pairs rc $? $(grep -h '"us"' $O/${i}_${v}_2048.json | tr -d ' \n')" | tee -a $O/timings.txt
done
date +%s > $O/t1
export FBK_POOL_MAX_BYTES=$((10 << 30))
(FBK_LIB_PATH=$R/build_variants/emitloop/libfbk.so timeout 230 python -m pytest tests -m gpu -v -n 8 --timeout 200 -p no:cacheprovider -rf > $O/pytest_emitloop_full.log 2>&1; echo "rc $?" >> $O/pytest_emitloop_full.log) &
(timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_struct.py tests/test_gpu_prepared.py tests/test_gpu_compact.py tests/test_gpu_queries.py \
   "tests/test_gpu_fullsize.py::test_config3_row_pairs_every_pair_vs_oracle" -m gpu -v -n 4 --timeout 180 -p no:cacheprovider -rf > $O/pytest_product_pairs.log 2>&1; echo "rc $?" >> $O/pytest_product_pairs.log) &
wait
date +%s > $O/t2
grep -c PASSED $O/pytest_emitloop_full.log; grep -E 'FAILED|ERROR' $O/pytest_emitloop_full.log | head -20; tail -4 $O/pytest_emitloop_full.log
grep -c PASSED $O/pytest_product_pairs.log; grep -E 'FAILED|ERROR' $O/pytest_product_pairs.log | head -20; tail -4 $O/pytest_product_pairs.log
