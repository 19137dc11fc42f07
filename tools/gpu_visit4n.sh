#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for E in 16384 131072; do
PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_probe.so timeout 600 python tools/phase_probe.py --envs $E --steps 400 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04n_phase_probe_$E.json | grep -E "row|ms_per|complex|chain|free|tests"
done
