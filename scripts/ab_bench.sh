#!/bin/bash
# A/B of the headline on ONE box (boxes of the pool differ by +-1 %): bash scripts/ab_bench.sh "ENV_A=1" "ENV_B=1" [reps]
# prints frames/s of alternating runs of `bench.py --steps 8 --warmup 2 --no-cpu-baseline --secondary none --no-roofline`
A="$1"; B="$2"; R=${3:-2}
run() { env $1 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --secondary none --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%s  %.2f frames/s  %.3f ms/step' % ('$1', d['value'], d['ms_per_step']))"; }
for i in $(seq $R); do run "$A"; run "$B"; done
