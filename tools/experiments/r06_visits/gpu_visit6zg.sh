#!/bin/bash
# (1) what the longest row wave of each step is made of (trace build of the PREVIOUS source state: clamp-free stages without the switch)
# (2) the switch from clamp-free to clamping stages near the bound: bit-identity against the build without it, kernel-duration distributions
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
timeout 900 python tools/wave_trace.py --envs 131072 --steps 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06zf_wave_trace.txt
timeout 900 python tools/ab_identity.py $C/libpbre.so $C/libpbre_free0.so 131072 1200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06zg_switch_ab.txt
for L in libpbre.so libpbre_free0.so; do
  echo "== $L (131072 envs)" | tee -a gpurun_out/r06zg_switch_ab.txt
  bash tools/trace_panda_steady3.sh 131072 r06zg PBRE_BENCH_NO_RT=1 PBRE_LIB=$C/$L 2>&1 | grep -E "min |span" | tee -a gpurun_out/r06zg_switch_ab.txt
done
echo "== libpbre.so (16384 envs)" | tee -a gpurun_out/r06zg_switch_ab.txt
bash tools/trace_panda_steady3.sh 16384 r06zg PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | tee -a gpurun_out/r06zg_switch_ab.txt
