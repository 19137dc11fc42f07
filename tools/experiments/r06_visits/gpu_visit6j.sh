#!/bin/bash
export TMPDIR=/tmp
PROBE_TORCH=0 timeout 600 python tools/host_async_probe.py 2>&1 | grep -vE "amdgpu.ids" | cut -c1-600
echo "== the failing RT tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "residual_threshold_contact_rich" 2>&1 | grep -vE "^/opt/amdgpu" | grep -E "Error|assert|flip|sweep|passed|failed" | head -20 | cut -c1-400
