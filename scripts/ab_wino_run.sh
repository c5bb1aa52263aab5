#!/bin/bash
# on the GPU box: every knock-out build of scripts/ab_wino.sh on the 3x3 layers of the step at 8 clips per call (geometry 43)
OUT=gpurun_out/${1:-wko}.log; shift
: > $OUT
for D in build/ab/wko_*; do
  echo "== ${D#build/ab/wko_}" >> $OUT
  WINO=1 ONLY="${ONLY:-x8}" ACCEL_LIB_PATH=$D/libaccel_hip.so timeout 300 python scripts/microbench/bench_conv.py ${TILES:-43} 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
