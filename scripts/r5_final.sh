# the last gpurun call of round 5 (10 GPU-minutes left): short measurements first, then the whole -m gpu suite on six workers
# (each step under its own timeout; everything lands in gpurun_out/r5z as it is produced)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5z
mkdir -p $O
cd $R
date +%s > $O/t0
# 1. the write path alone: what one-wavefront-per-8-KiB-cell kernels can reach (k_shift / k_flip / k_bsi_add / plain pair set-ops)
(timeout 60 scripts/write_rate > $O/write_rate.txt 2>&1; echo "rc $?" >> $O/write_rate.txt)
# 2. the checkers, built once before several test processes want them
python -c "from oracle import pyoracle; pyoracle.build()" > $O/oracle_build.log 2>&1
# 3. container slots per wave of the pair count on config 3's rows: 8192 row pairs (256 shards) and 2048 (64 shards)
V='pair_kernels=2,pair_spw=1;pair_kernels=2,pair_spw=2;pair_kernels=2,pair_spw=4;pair_spw=1,pair_kernels=2;pair_spw=2,pair_kernels=2'
(timeout 200 python scripts/bench_pairs.py --shards 256 --iters 20 --only-count --variants "$V" --out $O/pairs_spw_256.json > $O/pairs_spw_256.log 2>&1; echo "rc $?" >> $O/pairs_spw_256.log)
(timeout 60 python scripts/bench_pairs.py --shards 64 --iters 30 --only-count --variants "$V" --out $O/pairs_spw_64.json > $O/pairs_spw_64.log 2>&1; echo "rc $?" >> $O/pairs_spw_64.log)
date +%s > $O/t1
# 4. the whole GPU suite (the pools of six processes side by side: capped at 12 GiB each instead of an eighth of the device)
FBK_POOL_MAX_BYTES=$((12 << 30)) timeout ${PYTEST_TIMEOUT:-560} python -m pytest tests -m gpu -q -n 6 --timeout 500 -p no:cacheprovider -rf > $O/pytest.log 2>&1
echo "rc $?" >> $O/pytest.log
date +%s > $O/t2
(timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log)
tail -30 $O/pytest.log
cat $O/write_rate.txt
grep -h '"us"' $O/pairs_spw_256.json $O/pairs_spw_64.json
