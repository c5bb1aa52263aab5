#!/bin/bash
# Timing-only builds of libaccel_hip.so with -D switches (WRONG results by design), for same-box A/B through ACCEL_LIB_PATH:
#   bash scripts/ab_variants.sh name "-DFLAG1 -DFLAG2"   ->  build/ab/<name>/libaccel_hip.so
set -e
N=$1; F=$2
D=build/ab/$N
mkdir -p $D
for f in conv_igemm conv_b3r conv_b3d conv_wino conv_wino_b3 conv_wino_b3s conv_stem conv_stem_b3 conv_1x1ws conv_halo misc; do
  (hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $F -c accel_amd/csrc/$f.hip -o $D/$f.o) &
done
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w $F -x hip -c accel_amd/csrc/accel_hip.cpp -o $D/accel_hip.o &
wait
hipcc --offload-arch=gfx950 -shared -o $D/libaccel_hip.so $D/*.o -ldl
mkdir -p $D/tune && cp accel_amd/tune/gfx950.tune $D/tune/
ls -la $D/libaccel_hip.so
