#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
: > gpurun_out/r06zh_switch_ab.txt
for L in "$@"; do
  echo "== $L (131072 envs)" | tee -a gpurun_out/r06zh_switch_ab.txt
  bash tools/trace_panda_steady3.sh 131072 r06zh PBRE_BENCH_NO_RT=1 PBRE_LIB=$C/$L 2>&1 | grep -E "min |span" | tee -a gpurun_out/r06zh_switch_ab.txt
done
