#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04y_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04y_pytest_gpu.log | tail -6 | cut -c1-300
V=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_noguard.so
for r in 1 2; do
echo "--- default"; timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
echo "--- noguard"; PBRE_LIB=$V timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
done
