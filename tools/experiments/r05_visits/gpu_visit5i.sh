#!/bin/bash
# Round 5, visit I: phase probe of the default build (where does a row wave's time go beside the sweeps?)
export TMPDIR=/tmp
D=$(pwd)/pybullet-robot-envs_amd/csrc
for N in 16384 131072; do
  echo "--- probe $N"; PBRE_LIB=$D/libpbre_probe.so timeout 300 python tools/phase_probe.py --envs $N --steps 300 2>&1 | grep -v amdgpu
done | tee gpurun_out/r05i_phase_probe.txt
