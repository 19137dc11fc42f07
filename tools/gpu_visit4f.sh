#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_vec_env.py -q -x -m gpu > gpurun_out/r04f_pytest.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04f_pytest.log | tail -6 | cut -c1-300
timeout 600 python tools/tail_probe.py --sizes 4096,16384,32768,65536,131072 --preroll 1100 2>&1 | grep "^{" | tee gpurun_out/r04f_shards.json | cut -c1-330
PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_nocas.so timeout 600 python tools/tail_probe.py --sizes 16384,65536,131072 --preroll 1100 2>&1 | grep "^{" | sed "s/^{/{\"lib\": \"before (nocas build of the previous commit)\", /" | tee -a gpurun_out/r04f_shards.json | cut -c1-330
