#!/bin/bash
# same-box A/B of two launch-geometry tables: bash scripts/ab_tune.sh old.tune new.tune [reps]
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for T in $A $B; do
    cp $T /tmp/ab_table.tune      # (a copy: the library appends decisions it had to time)
    for batch in 8 1; do
      ACCEL_TUNE_SHIPPED=0 ACCEL_TUNE_CACHE=/tmp/ab_table.tune python bench.py --batch $batch --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --secondary none 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T batch $batch: %.1f frames/s  %.3f ms' % (d['value'], d['ms_per_step']))"
    done
  done
done
