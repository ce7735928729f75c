# one gpurun call: the pair kernels of several library builds side by side on config 3's 8192 row pairs
#   LIBS="base xcd nt xcdnt base" bash scripts/ab_pairs.sh        (base = the product library; others = build_variants/<name>/libfbk.so)
R=$GRAFT_REPO_ROOT
TAG=${TAG:-ab1}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
i=0
for v in ${LIBS:-base}; do
  i=$((i + 1))
  lib=""
  [ "$v" != base ] && lib=$R/build_variants/$v/libfbk.so
  FBK_LIB_PATH=$lib timeout 60 python scripts/bench_pairs.py --shards ${SHARDS:-256} --iters ${ITERS:-20} --ops "${OPS:-intersectionCount,intersect + optimize()}" \
    --variants "pair_kernels=2" ${EXTRA:-} --out $O/${i}_$v.json > $O/${i}_$v.log 2>&1
  echo "$i $v rc $? $(grep -h '"us"' $O/${i}_$v.json | tr -d ' \n')"
done
