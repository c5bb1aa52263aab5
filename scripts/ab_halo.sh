#!/bin/bash
# Knock-out builds of conv_halo.hip (timing only, WRONG results): bash scripts/ab_halo.sh BASE NO_MMA ... -> build/ab/hko_<NAME>/libaccel_hip.so
set -e
for N in "$@"; do
  D=build/ab/hko_$N
  mkdir -p $D/tune && cp accel_amd/tune/gfx950.tune $D/tune/
  ( F=""; [ "$N" != "BASE" ] && F="-DHKO_$N"; [ "$N" = "SGB" ] && F="-DHALO_SGB"
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $F -c accel_amd/csrc/conv_halo.hip -o $D/conv_halo.o
    hipcc --offload-arch=gfx950 -shared -o $D/libaccel_hip.so $D/conv_halo.o $(ls accel_amd/csrc/*.o | grep -v conv_halo.o) -ldl ) &
done
wait
ls build/ab/hko_*/libaccel_hip.so
