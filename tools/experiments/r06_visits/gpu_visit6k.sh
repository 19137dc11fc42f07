#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
PROBE_TORCH=0 timeout 600 python tools/host_async_probe.py 2>&1 | grep -vE "amdgpu.ids" | cut -c1-600 | tee gpurun_out/r06k_host_async_probe.txt
PROBE_TORCH=0 PBRE_ASYNC_D2H=0 timeout 600 python tools/host_async_probe.py 2>&1 | grep -E "pipelined" | cut -c1-300 | tee -a gpurun_out/r06k_host_async_probe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pipelined_host" 2>&1 | grep -vE "^/opt/amdgpu" | tail -2 | cut -c1-300
echo "== the failing RT tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "residual_threshold_contact_rich" 2>&1 | grep -vE "^/opt/amdgpu" | grep -B2 -A12 "^E " | head -60 | cut -c1-500
