#!/bin/bash
# PMC passes over the Winograd microbench of one launch geometry (default 41): where do the wavefronts of the kernel wait?
#   bash scripts/debug/wb3_pmc.sh [tile]      (on the GPU box; output under gpurun_out/wb3_pmc/)
set -u
T=${1:-41}
REPO=$(pwd)
OUT=$REPO/gpurun_out/wb3_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
CMD="python $REPO/scripts/microbench/bench_conv.py $T"
WINO=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/p1 -o w --output-format csv -- $CMD > $OUT/p1.log 2>&1
WINO=1 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d $OUT/p2 -o w --output-format csv -- $CMD > $OUT/p2.log 2>&1
# (a third pass with TCP_* / TA_* counters hung the profiler on this stack for the full time limit: not collected)
cd $REPO
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("OUT", "gpurun_out/wb3_pmc")
for p in ("p1", "p2"):
    fs = glob.glob("%s/%s/**/*counter_collection.csv" % (out, p), recursive=True)
    if not fs:
        print(p, "no csv;", open("%s/%s.log" % (out, p)).read()[-600:])
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = (r["Kernel_Name"][:40], r["Grid_Size"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "wino" not in k[0]: continue
        print(p, k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
