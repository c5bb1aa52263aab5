#!/bin/bash
# PMC diagnosis of one conv shape (separate passes, counters only): bash scripts/microbench/pmc_one_conv.sh <tag> <tile> cin cout H W k
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc1_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  ITERS=40 timeout 120 rocprofv3 --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python $REPO/scripts/microbench/one_conv.py "$@" > /dev/null 2>&1
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if not any(k in r["Kernel_Name"] for k in ("conv_igemm", "conv_wino", "conv_stem", "conv1x1_ws")): continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc): print("%-28s %16.0f per launch" % (k, acc[k] / max(len(n[k]), 1)))
PY
