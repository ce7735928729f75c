#!/bin/bash
# rocprofv3 --pmc WRITE_SIZE and FETCH_SIZE over a driver script: HBM bytes per kernel
#   scripts/pmc_write_misc.sh <out dir under gpurun_out> [driver.py=profile_misc.py] [args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; mkdir -p $O
DRV=${2:-profile_misc.py}; shift; shift
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o b -- python $R/scripts/$DRV "$@" > $O/misc_w.json 2> $O/w.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o b -- python $R/scripts/$DRV "$@" > $O/misc_f.json 2> $O/f.err
python3 $R/scripts/pmc_hbm_summary.py $O "python scripts/$DRV $*"; cat $O/misc_w.json
