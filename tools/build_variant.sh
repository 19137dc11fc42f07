#!/bin/bash
# Build an A/B variant of libpbre.so with extra compiler defines:
#   tools/build_variant.sh <suffix> "<-DFLAG=...>" [tu]   -> csrc/libpbre_<suffix>.so
# tu: the translation unit that is recompiled with the flags -- pbre_capi (default: the Panda engine; compiled as ONE unit with
# -DPBRE_UNITY, i.e. with every (MODE, RT) instantiation of the step launcher inside, ~10 min), pbre_step_<m>_<rt> (one instantiation
# of the launcher, e.g. pbre_step_2_false = env.step() under joint control: ~4 min), pbre_lane (the iCub's lane-per-env engine), ...;
# the others are linked from obj/ (build.sh first).  Run with PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_<suffix>.so
set -e
cd "$(dirname "$0")/../pybullet-robot-envs_amd/csrc"
SUF=$1; FL=$2; TU=${3:-pbre_capi}
EXTRA=""; SRC=$TU.hip
[ $TU = pbre_lane ] && EXTRA="-mllvm -pragma-unroll-threshold=1000000"
[ $TU = pbre_capi ] && EXTRA="-DPBRE_UNITY"
case $TU in pbre_step_*) m=${TU#pbre_step_}; EXTRA="-DPBRE_INST_MODE=${m%%_*} -DPBRE_INST_RT=${m#*_}"; SRC=pbre_step_inst.hip;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage $EXTRA $FL -c -o obj/${TU}_$SUF.o $SRC 2> obj/${TU}_$SUF.log
OBJS=""
STEPS=""; for rt in true false; do for m in 0 1 2 3 4 5; do STEPS="$STEPS pbre_step_${m}_${rt}"; done; done
[ $TU = pbre_capi ] && STEPS=""
for t in $STEPS pbre_capi pbre_wide pbre_hands pbre_lane pbre_icub_arm pbre_comm; do
    if [ $t = $TU ]; then OBJS="$OBJS obj/${TU}_$SUF.o"; else OBJS="$OBJS obj/$t.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libpbre_$SUF.so $OBJS -ldl
echo built libpbre_$SUF.so
