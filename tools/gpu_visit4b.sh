#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q > gpurun_out/r04b_pytest_parity.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04b_pytest_parity.log | tail -12 | cut -c1-300
PBRE_PARITY_MEASURE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "reset_and_steps or free_running or ragged or force_limited or config2" 2>&1 | grep -E "MEASURED \(reset|MEASURED \(60|MEASURED \(force|MEASURED \(config|passed|failed" | sort | uniq -c | sort -rn | head -40 | cut -c1-700 > gpurun_out/r04b_measured.log; tail -3 gpurun_out/r04b_measured.log | cut -c1-200
