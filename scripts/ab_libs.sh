#!/bin/bash
# same-box A/B of builds of the library: bash scripts/ab_libs.sh <reps> <dir with libaccel_hip.so + tune/> ...   (per-op lines matching $OPS + the headline)
R=$1; shift
for r in $(seq $R); do for D in "$@"; do
  echo "== $D"
  ACCEL_LIB_PATH=$D/libaccel_hip.so python scripts/microbench/prof_ops.py 18 8 2>/dev/null | grep -iE "${OPS:-key plan|cur plan}" | cut -c1-100
  ACCEL_LIB_PATH=$D/libaccel_hip.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --secondary none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench %.1f frames/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
