#!/bin/bash
# Knock-out builds of conv_wino_b3s.hip (timing only, WRONG results): one library per -D switch, the other objects from the tree's
# own build.  bash scripts/ab_wino.sh NO_MMA NO_SPLIT ...   ->  build/ab/wko_<NAME>/libaccel_hip.so (select with ACCEL_LIB_PATH)
set -e
for N in "$@"; do
  D=build/ab/wko_$N
  mkdir -p $D/tune && cp accel_amd/tune/gfx950.tune $D/tune/
  ( F=""; [ "$N" != "BASE" ] && F="-DWKO_$N"
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $F -c accel_amd/csrc/conv_wino_b3s.hip -o $D/conv_wino_b3s.o
    hipcc --offload-arch=gfx950 -shared -o $D/libaccel_hip.so $D/conv_wino_b3s.o $(ls accel_amd/csrc/*.o | grep -v conv_wino_b3s.o) -ldl ) &
done
wait
ls -la build/ab/wko_*/libaccel_hip.so
