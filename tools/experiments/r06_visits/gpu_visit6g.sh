#!/bin/bash
# Round 6, visit G: the contention tests with the harness race fixed (5 runs); where the pipelined host path's time goes
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_contention.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | grep -E "rows differ|passed|failed" | cut -c1-300; done
for V in 1 0; do PBRE_ASYNC_D2H=$V timeout 600 python tools/host_async_probe.py 2>&1 | grep -vE "amdgpu.ids"; done | tee gpurun_out/r06g_host_async_probe.txt
