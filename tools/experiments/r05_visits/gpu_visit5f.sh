#!/bin/bash
# Round 5, visit F: every GPU test (incl. floating-base iCub, tightened closed-loop push), the self-collision frequency probe.
TAG=${1:-r05f}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -25 | cut -c1-300
echo "== self-collision probe"
timeout 600 python tools/self_collision_probe.py --envs 16384 --samples 12 2>&1 | tail -1 | tee gpurun_out/${TAG}_self_collision_probe.json | cut -c1-700
