#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04u_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04u_pytest_gpu.log | tail -8 | cut -c1-300
timeout 600 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep "^{" | tee gpurun_out/r04u_shards.json | cut -c1-330
