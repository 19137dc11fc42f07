#!/bin/bash
# Builds libpbre.so (HIP engine, gfx950) in-tree.  Cross-compiles without a GPU.  The six translation units are compiled
# in parallel and only when their sources changed (obj/ is scratch).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
mkdir -p obj
HDRS="pbre_math.hpp pbre_sidepick.hpp pbre_core.hpp pbre_objstep.hpp pbre_fast.hpp pbre_lane.hpp pbre_host.hpp pbre_tables.hpp pbre_wide.hpp pbre_wide_impl.hpp lanes_device.hpp pbre_comm_impl.hpp ../../include/pbre.h build.sh"
pids=()
for tu in pbre_capi pbre_wide pbre_hands pbre_lane pbre_icub_arm pbre_comm; do
    stale=0
    [ -f obj/$tu.o ] || stale=1
    for f in $tu.hip $HDRS; do [ $stale = 1 ] || [ obj/$tu.o -nt $f ] || stale=1; done
    if [ $stale = 1 ]; then
        extra=""
        [ $tu = pbre_lane ] && extra="-mllvm -pragma-unroll-threshold=1000000"     # the fully unrolled 20-link code of pbre_lane.hpp
        ( $HIPCC $FLAGS $extra -c -o obj/$tu.o.tmp $tu.hip 2> obj/$tu.log && mv obj/$tu.o.tmp obj/$tu.o ) &
        pids+=($!)
    fi
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
cat obj/pbre_capi.log obj/pbre_wide.log obj/pbre_hands.log obj/pbre_lane.log obj/pbre_icub_arm.log obj/pbre_comm.log > build.log 2>/dev/null || true
if [ $rc != 0 ]; then grep -E "error|Error" -A3 build.log | head -60; exit 1; fi
$HIPCC --offload-arch=gfx950 -fPIC -shared -o libpbre.so obj/pbre_capi.o obj/pbre_wide.o obj/pbre_hands.o obj/pbre_lane.o obj/pbre_icub_arm.o obj/pbre_comm.o -ldl
grep -E "Name:|VGPRs:|ScratchSize|Occupancy" build.log | sed -E "s/.*(Name: [^ ]+|VGPRs: [0-9]+|ScratchSize[^:]*: [0-9]+|Occupancy[^:]*: [0-9]+).*/\1/" | paste - - - - | grep -E "k_step|k_fast|kw_step|kw_ik|kw_lane|kw_list" || true
