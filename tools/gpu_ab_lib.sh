#!/bin/bash
# same-box A/B of libpbre.so against a build variant (tools/build_variant.sh):  tools/gpu_ab_lib.sh <suffix> [sizes] [repeats]
S=$1; SIZES=${2:-16384,131072}; R=${3:-2}
export TMPDIR=/tmp
V=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_$S.so
for r in $(seq $R); do
echo "--- default"; timeout 300 python tools/tail_probe.py --sizes $SIZES --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
echo "--- $S"; PBRE_LIB=$V timeout 300 python tools/tail_probe.py --sizes $SIZES --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
done
