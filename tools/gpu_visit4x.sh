#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04x_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04x_pytest_gpu.log | tail -8 | cut -c1-300
timeout 300 python tools/tail_probe.py --sizes 16384,65536,131072 --preroll 1100 --steps 600 2>&1 | tail -12
timeout 300 python bench.py --no-shards 2>&1 | tail -1 | cut -c1-1500
