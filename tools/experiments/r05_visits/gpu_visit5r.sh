#!/bin/bash
# Round 5, last visit: the complete GPU test suite and smoke() on the final build
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r05_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r05_pytest_gpu.log | tail -4 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r05_smoke.log
