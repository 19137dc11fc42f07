#!/bin/bash
# Builds libpbre.so (HIP engine, gfx950) in-tree.  Cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-fast-math -fno-slp-vectorize \
    -Rpass-analysis=kernel-resource-usage \
    -o libpbre.so pbre_capi.hip pbre_wide.hip 2> build.log || { cat build.log; exit 1; }
grep -E "Name:|VGPRs:|ScratchSize|Occupancy" build.log | sed -E "s/.*(Name: [^ ]+|VGPRs: [0-9]+|ScratchSize[^:]*: [0-9]+|Occupancy[^:]*: [0-9]+).*/\1/" | paste - - - - | grep -E "k_step|k_fast|kw_step|kw_ik" || true
