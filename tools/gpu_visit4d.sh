#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r04d_pytest_parity.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04d_pytest_parity.log | tail -4 | cut -c1-300
timeout 600 python tools/tail_probe.py --sizes 4096,16384,32768,65536,131072 --preroll 1100 2>&1 | grep "^{" | tee gpurun_out/r04d_shards.json | cut -c1-330
