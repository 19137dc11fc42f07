#!/bin/bash
# Build an A/B variant of libpbre.so with extra compiler defines:
#   tools/build_variant.sh <suffix> "<-DFLAG=...>" [tu]   -> csrc/libpbre_<suffix>.so
# tu: the translation unit that is recompiled with the flags -- pbre_capi (default, the Panda engine) or pbre_lane (the iCub's
# lane-per-env engine); the others are linked from obj/.  Run with PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_<suffix>.so
set -e
cd "$(dirname "$0")/../pybullet-robot-envs_amd/csrc"
SUF=$1; FL=$2; TU=${3:-pbre_capi}
EXTRA=""
[ $TU = pbre_lane ] && EXTRA="-mllvm -pragma-unroll-threshold=1000000"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize $EXTRA $FL -c -o obj/${TU}_$SUF.o $TU.hip
OBJS=""
for t in pbre_capi pbre_wide pbre_hands pbre_lane pbre_icub_arm pbre_comm; do
    if [ $t = $TU ]; then OBJS="$OBJS obj/${TU}_$SUF.o"; else OBJS="$OBJS obj/$t.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libpbre_$SUF.so $OBJS -ldl
echo built libpbre_$SUF.so
