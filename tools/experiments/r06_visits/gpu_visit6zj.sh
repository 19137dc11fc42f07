#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
: > gpurun_out/r06zj_switch_ab.txt
for L in libpbre_sw2.so libpbre_sw0.so; do
  echo "== $L (16384 envs)" | tee -a gpurun_out/r06zj_switch_ab.txt
  bash tools/trace_panda_steady3.sh 16384 r06zj PBRE_BENCH_NO_RT=1 PBRE_LIB=$C/$L 2>&1 | grep -E "min |span" | tee -a gpurun_out/r06zj_switch_ab.txt
done
for N in 131072 16384; do timeout 900 python tools/ab_identity.py $C/libpbre_sw2.so $C/libpbre_sw0.so $N 1200 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06zj_switch_ab.txt; done
