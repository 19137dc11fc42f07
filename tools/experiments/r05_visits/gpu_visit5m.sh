#!/bin/bash
# Round 5, visit M: phase probe of the row waves under the one-launch step (PBRE_FUSED=1) and under the two-kernel step (0): same probe build
export TMPDIR=/tmp
D=$(pwd)/pybullet-robot-envs_amd/csrc
for N in 16384 131072; do for V in 1 0; do
  echo "--- probe $N PBRE_FUSED=$V"; PBRE_FUSED=$V PBRE_LIB=$D/libpbre_probe.so timeout 300 python tools/phase_probe.py --envs $N --steps 300 2>&1 | grep -v amdgpu
done; done | tee gpurun_out/r05m_phase_probe.txt
