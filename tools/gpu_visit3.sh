#!/bin/bash
TAG=${1:-r02c}
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -40 | cut -c1-400
echo "== complex breakdown"
timeout 300 python tools/complex_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -8
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
echo "== bench short"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | cut -c1-1400
