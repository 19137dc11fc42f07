#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/tail_probe.py --sizes 4096,16384,32768,65536,131072 --preroll 1100 2>&1 | grep "^{" | tee gpurun_out/r04g_shards.json | cut -c1-330
for V in probe probe_a; do
PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_$V.so timeout 600 python tools/phase_probe.py --envs 16384 2>&1 | grep -v amdgpu.ids | grep -E "row|ms_per|complex" | tee gpurun_out/r04g_phase_probe_$V.json
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_vec_env.py -q -x -m gpu > gpurun_out/r04g_pytest.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04g_pytest.log | tail -6 | cut -c1-300
