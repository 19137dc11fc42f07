#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/valu_bisect.sh main 2>&1 | grep -v "^W2026"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04z_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04z_pytest_gpu.log | tail -6 | cut -c1-300
timeout 300 python tools/tail_probe.py --sizes 4096,16384,65536,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
timeout 300 python bench.py --no-shards --no-other-configs 2>&1 | tail -1 | cut -c1-300
for M in "" "--joint"; do PBRE_ICUB_LANE=1 timeout 300 python tools/bench_icub.py --envs 32768 --steps 20 $M 2>&1 | tail -1 | cut -c1-200; done
