#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun; outputs under gpurun_out/, summaries are then copied
# to profiles/ by hand):  bash scripts/profile_round.sh r01
# 1. plain bench (headline + secondaries)
# 2. rocprofv3 --kernel-trace --stats of the same command (plans run on one stream: per-launch durations are exact)
# 3. three separate --pmc passes (SQ / FETCH_SIZE / WRITE_SIZE), never combined with the trace domains
# 4. per-op HIP-event timings of both plans
set -u
R=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
# launch geometries come from the shipped table (accel_amd/tune/gfx950.tune): no tuning launches in any of the runs below
python bench.py > $OUT/bench_${R}_final.json 2> $OUT/bench_${R}_final.err
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --secondary none"
rocprofv3 --kernel-trace --stats -d $OUT/prof_$R -o bench --output-format csv -- $B > $OUT/prof_${R}_bench.json 2>/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/prof_${R}_pmc1 -o bench --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_${R}_pmc2 -o bench --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_${R}_pmc3 -o bench --output-format csv -- $B > /dev/null 2>&1
cd $REPO
python scripts/microbench/prof_ops.py 18 > $OUT/ops18_${R}.txt 2>&1
python scripts/microbench/prof_ops.py 18 8 > $OUT/ops18_${R}_b8.txt 2>&1
python scripts/summarize_rocprof.py $OUT/prof_$R $OUT/prof_${R}_pmc1 $OUT/prof_${R}_pmc2 $OUT/prof_${R}_pmc3 $OUT/pmc_traffic_$R.json batch=8 > $OUT/rocprof_summary_$R.md
tail -3 $OUT/bench_${R}_final.json | cut -c1-300
