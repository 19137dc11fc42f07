#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04k_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04k_pytest_gpu.log | tail -8 | cut -c1-300
