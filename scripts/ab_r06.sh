#!/bin/bash
# same-box A/B of two builds of the library, headline (8 clips per call) and one clip per call:
#   bash scripts/ab_r06.sh <reps> <dir with libaccel_hip.so + tune/> ...      ("." = this tree)
R=$1; shift
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %8.2f frames/s  %8.3f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
ARGS="--warmup 2 --no-cpu-baseline --secondary none --no-roofline"
for i in $(seq $R); do for D in "$@"; do
  L=$D/libaccel_hip.so; [ "$D" = "." ] && L=$PWD/accel_amd/libaccel_hip.so
  ACCEL_LIB_PATH=$L python bench.py --steps 10 $ARGS 2>/dev/null | one "$D batch8"
  ACCEL_LIB_PATH=$L python bench.py --steps 40 --batch 1 $ARGS 2>/dev/null | one "$D batch1"
done; done
