cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3q
timeout 300 python scripts/bench_ctops.py --rows 1024 --iters 20 --out gpurun_out/r3q/ctops_old.json --only Ary1,Ary1024,Run256,Ary4096 --opt pair_kernels=1 > gpurun_out/r3q/old.txt 2>&1
timeout 300 python scripts/bench_ctops.py --rows 1024 --iters 20 --out gpurun_out/r3q/ctops_new.json --only Ary1,Ary1024,Run256,Ary4096 --opt pair_kernels=2 > gpurun_out/r3q/new.txt 2>&1
