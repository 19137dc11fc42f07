#!/bin/bash
# Round 5, visit O: the new many-trips test of the one-launch step, then a soak of the default path (k_fused): 30 000 steps at 131072 envs
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch" 2>&1 | grep -vE "^/opt/amdgpu" | tail -4
SOAK_STEPS=${SOAK_STEPS:-30000} timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_soak.json | cut -c1-300
