#!/bin/bash
# same-box A/B of several build variants of libpbre.so (tools/build_variant.sh):  tools/gpu_ab_multi.sh "<suffix> <suffix> ..." [sizes] [repeats] [tag]
# ("default" = libpbre.so itself); interleaved, so that clock drift of the box hits every variant alike.  Log -> gpurun_out/<tag>_ab.txt
VARS=$1; SIZES=${2:-16384,131072}; R=${3:-2}; TAG=${4:-ab}
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in $(seq $R); do
  for S in $VARS; do
    if [ $S = default ]; then unset PBRE_LIB; else export PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_$S.so; fi
    echo "--- $S"; timeout 300 python tools/tail_probe.py --sizes $SIZES --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
  done
done 2>&1 | tee gpurun_out/${TAG}_ab.txt
