set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2t
mkdir -p $O
cd $R
(timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/pytest.log
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python bench.py --gpus 2 --steps 100 > $O/bench_n2.json 2> $O/bench_n2.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $R/bench.py --steps 200 --repeats 5 --no-cpu-baseline > $O/bench_prof.json 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python $R/bench.py --steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --cold-sets 1 --shards4 128 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python $R/bench.py --steps 20 --warmup 2 --repeats 2 --no-cpu-baseline --cold-sets 1 --shards4 128 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/misc -o misc -- python $R/scripts/profile_misc.py 64 > $O/misc.json 2> /dev/null
cd $R
timeout 120 python scripts/bsi_bench.py 2>&1 | grep -v amdgpu.ids > $O/bsi_bench.txt
if [ "$CTOPS" = 1 ]; then timeout 300 python scripts/bench_ctops.py --rows 1024 --iters 20 --out $O/ctops.json --only Ary1,Ary16,Ary256,BM4096,Run16,Run1024 2>&1 | grep -v amdgpu.ids | tail -30 > $O/ctops_small.txt; fi
ls $O $O/kt $O/pmc_fetch | head -30
tail -3 $O/pytest.log
