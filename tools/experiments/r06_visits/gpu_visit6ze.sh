#!/bin/bash
# default step: what shapes the upper tail of k_fused's duration?  FREE motor stages off (no wave ever starts over; chain ~17 % longer)
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
: > gpurun_out/r06ze_tail_ab.txt
for L in libpbre.so libpbre_free0.so; do
  echo "== $L (131072 envs)" | tee -a gpurun_out/r06ze_tail_ab.txt
  bash tools/trace_panda_steady3.sh 131072 r06ze PBRE_BENCH_NO_RT=1 PBRE_LIB=$C/$L 2>&1 | grep -E "min |span" | tee -a gpurun_out/r06ze_tail_ab.txt
done
