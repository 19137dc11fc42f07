#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/tail_probe.py --sizes 4096,16384,32768,65536,131072 --preroll 1100 2>&1 | grep "^{" | tee gpurun_out/r04h_shards.json | cut -c1-330
PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_nocas.so timeout 600 python tools/tail_probe.py --sizes 16384,65536,131072 --preroll 1100 2>&1 | grep "^{" | sed "s/^{/{\"lib\": \"round-3 row kernel\", /" | tee -a gpurun_out/r04h_shards.json | cut -c1-330
for V in probe; do
PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_$V.so timeout 600 python tools/phase_probe.py --envs 16384 2>&1 | grep -v amdgpu.ids | grep -E "row|ms_per|complex" | tee gpurun_out/r04h_phase_probe_$V.json
done
