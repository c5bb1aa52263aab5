#!/bin/bash
# Regenerates the shipped launch-geometry table accel_amd/tune/gfx950.tune on a GPU box (through gpurun):
#   bash scripts/make_tune_table.sh   -> gpurun_out/gfx950.tune, to be copied to accel_amd/tune/gfx950.tune
# Every workload the BASELINE configs and bench.py bind is bound once with timing-based selection; the decisions are
# then replayed by every later process (accel_hip.cpp, "Launch-geometry decisions are persisted").
set -u
REPO=$(pwd)
export ACCEL_TUNE_SHIPPED=0 ACCEL_TUNE_CACHE=$REPO/gpurun_out/gfx950.tune
rm -f $ACCEL_TUNE_CACHE
if [ "${NEW_ONLY:-0}" = "1" ]; then
  # a new launch geometry changed the key of a few layers (e.g. the bf16x3 stem kernel: the stems' keys carry one more bit): start
  # from the shipped table, bind every workload once -- only the layers whose key is not in the table are timed and appended
  cp accel_amd/tune/gfx950.tune $ACCEL_TUNE_CACHE
fi
if [ "${F16_ONLY:-0}" = "1" ]; then
  # only the fp16-mode layers are re-timed (a new f16 launch geometry was added): the fp32 lines of the shipped table stay
  python bench.py --version 50 --dtype f16 --size 2048x4096 --interval 10 --batch 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --secondary none > /dev/null 2>> gpurun_out/tune_table.err
  python - "$ACCEL_TUNE_CACHE" accel_amd/tune/gfx950.tune <<'PY'
import sys
new, old = sys.argv[1], sys.argv[2]
is_f16 = lambda l: not l.startswith("#") and len(l.split()) > 14 and int(l.split()[14]) & 16
head = open(new).readline()
keep = [l for l in open(old) if not l.startswith("#") and not is_f16(l)]
add = [l for l in open(new) if is_f16(l)]
open(new, "w").write(head + "".join(keep) + "".join(add))
print("kept %d fp32 lines, %d f16 lines re-timed" % (len(keep), len(add)))
PY
  wc -l $ACCEL_TUNE_CACHE
  exit 0
fi
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> gpurun_out/tune_table.err          # Accel-18 B=8, B=1, Accel-101 B=8, and B=8 with ACCEL_WITHHOLD=split
for v in 34 50 101; do
  python bench.py --version $v --batch 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --secondary none > /dev/null 2>> gpurun_out/tune_table.err
done
for v in 34 50; do
  python bench.py --version $v --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --secondary none > /dev/null 2>> gpurun_out/tune_table.err
done
python bench.py --size 512x1024 --batch 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --secondary none > /dev/null 2>> gpurun_out/tune_table.err
python bench.py --version 50 --dtype f16 --size 2048x4096 --interval 10 --batch 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --secondary none > /dev/null 2>> gpurun_out/tune_table.err    # config 5
wc -l $ACCEL_TUNE_CACHE
