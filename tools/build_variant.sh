#!/bin/bash
# Build an A/B variant of libpbre.so with extra compiler defines: tools/build_variant.sh <suffix> "<-DFLAG=...>" -> csrc/libpbre_<suffix>.so
# (only pbre_capi.hip -- the Panda engine -- is recompiled; run with PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_<suffix>.so)
set -e
cd "$(dirname "$0")/../pybullet-robot-envs_amd/csrc"
SUF=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize "$@" -c -o obj/pbre_capi_$SUF.o pbre_capi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libpbre_$SUF.so obj/pbre_capi_$SUF.o obj/pbre_wide.o obj/pbre_hands.o
echo built libpbre_$SUF.so
