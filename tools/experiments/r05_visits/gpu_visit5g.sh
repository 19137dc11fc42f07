#!/bin/bash
# Round 5, visit G: phase probe of the row waves with and without the clamp-free motor stages (same box)
export TMPDIR=/tmp
D=$(pwd)/pybullet-robot-envs_amd/csrc
for V in probe0 probe1; do for N in 16384 131072; do
  echo "--- $V $N"; PBRE_LIB=$D/libpbre_$V.so timeout 300 python tools/phase_probe.py --envs $N --steps 300 2>&1 | grep -E "sweeps|paths|waves_per_step|restart|ms_per_step|row_wave_ticks" 
done; done | tee gpurun_out/r05g_phase_probe.txt
