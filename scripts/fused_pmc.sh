#!/bin/bash
# rocprofv3 --pmc passes over the in-kernel-decode count matrix (scripts/fused_pmc.py): SQ counters per launch.
#   scripts/fused_pmc.sh <out dir under gpurun_out> [shards=256] [ablate=0] [driver=fused_pmc.py] [kernel name filter]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; N=${2:-256}; A=${3:-0}; DRV=${4:-fused_pmc.py}; export KFILTER=${5:-count_matrix}
mkdir -p $O
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o b -- python $R/scripts/$DRV $N $A > $O/p$i.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, os, sys, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ["KFILTER"] in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
