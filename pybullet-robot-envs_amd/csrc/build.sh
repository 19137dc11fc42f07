#!/bin/bash
# Builds libpbre.so (HIP engine, gfx950) in-tree.  Cross-compiles without a GPU.  Translation units are compiled in parallel (at most
# $PBRE_BUILD_JOBS at a time, default: the core count) and only when their sources changed (obj/ is scratch).  The Panda engine's step
# launcher is instantiated once per (MODE, RT) in a translation unit of its own (pbre_step_inst.hip, round 6): as one unit it took 9-12 min.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage $PBRE_EXTRA_FLAGS"
JOBS=${PBRE_BUILD_JOBS:-$(nproc)}
mkdir -p obj
HDRS="pbre_math.hpp pbre_sidepick.hpp pbre_core.hpp pbre_objstep.hpp pbre_fast.hpp pbre_lane.hpp pbre_host.hpp pbre_tables.hpp pbre_wide.hpp pbre_wide_impl.hpp lanes_device.hpp pbre_comm_impl.hpp pbre_panda.hpp ../../include/pbre.h build.sh"
# name:source:extra flags   (the residual-exit instantiations first: they are the longest)
UNITS=""
for rt in true false; do for m in 2 0 3 1 4 5; do UNITS="$UNITS pbre_step_${m}_${rt}:pbre_step_inst.hip:-DPBRE_INST_MODE=${m}@-DPBRE_INST_RT=${rt}"; done; done
UNITS="$UNITS pbre_lane:pbre_lane.hip:-mllvm@-pragma-unroll-threshold=1000000"      # the fully unrolled 20-link code of pbre_lane.hpp
for tu in pbre_capi pbre_wide pbre_hands pbre_icub_arm pbre_comm; do UNITS="$UNITS $tu:$tu.hip:"; done
OBJS=""; LOGS=""
running=0; rc=0
for u in $UNITS; do
    name=${u%%:*}; rest=${u#*:}; src=${rest%%:*}; extra=$(echo "${rest#*:}" | tr '@' ' ')
    OBJS="$OBJS obj/$name.o"; LOGS="$LOGS obj/$name.log"
    stale=0
    [ -f obj/$name.o ] || stale=1
    # the unit's own headers (host-side preprocessor pass, < 0.1 s), not the whole list: a change to pbre_lane.hpp rebuilds pbre_lane.hip only
    deps=$($HIPCC --offload-arch=gfx950 -std=c++17 --cuda-host-only -MM $extra $src 2>/dev/null | tr ' \\' '\n\n' | grep -E "^[A-Za-z_./]+\.(hpp|h|hip)$" | grep -v "^/" | sort -u | tr '\n' ' ')
    [ -n "$deps" ] || deps="$src $HDRS"
    for f in $deps build.sh; do [ $stale = 1 ] || [ obj/$name.o -nt $f ] || stale=1; done
    if [ $stale = 1 ]; then
        while [ $running -ge $JOBS ]; do wait -n || rc=1; running=$((running - 1)); done
        ( $HIPCC $FLAGS $extra -c -o obj/$name.o.tmp $src 2> obj/$name.log && mv obj/$name.o.tmp obj/$name.o ) &
        running=$((running + 1))
    fi
done
while [ $running -gt 0 ]; do wait -n || rc=1; running=$((running - 1)); done
cat $LOGS > build.log 2>/dev/null || true
if [ $rc != 0 ]; then grep -E "error|Error" -A3 build.log | head -60; exit 1; fi
$HIPCC --offload-arch=gfx950 -fPIC -shared -o libpbre.so $OBJS -ldl
grep -E "Name:|VGPRs:|ScratchSize|Occupancy" build.log | sed -E "s/.*(Name: [^ ]+|VGPRs: [0-9]+|ScratchSize[^:]*: [0-9]+|Occupancy[^:]*: [0-9]+).*/\1/" | paste - - - - | grep -E "k_step|k_fast|k_fused|kw_step|kw_ik|kw_lane|kw_list" || true
