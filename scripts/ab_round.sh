#!/bin/bash
# same-box A/B of the headline: the round-4 tree (build/r04tree, a git worktree of 705566d) against this tree and its timing variants
#   bash scripts/ab_round.sh <reps> [variant ...]      variant = a directory under build/ab (scripts/ab_variants.sh) or ENV=value
R=${1:-2}; shift
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s %.2f frames/s  %.3f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --secondary none --no-roofline"
for i in $(seq $R); do
  [ -d build/r04tree ] && (cd build/r04tree && python bench.py $ARGS 2>/dev/null) | one r04
  python bench.py $ARGS 2>/dev/null | one now
  for v in "$@"; do
    case "$v" in
      *=*) env $v python bench.py $ARGS 2>/dev/null | one $v ;;
      *) ACCEL_LIB_PATH=$PWD/build/ab/$v/libaccel_hip.so python bench.py $ARGS 2>/dev/null | one $v ;;
    esac
  done
done
