#!/bin/bash
# Round 6, visit B: the extended VALU micro-benchmark; is test_icub_crafted_contact_states[1] flaky or a regression; full GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ubench/valu_rate.sh > /dev/null 2>&1
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_icub.py -m gpu -q -x -k "crafted_contact_states" 2>&1 | grep -vE "^/opt/amdgpu" | tail -3 | cut -c1-400; done
echo "== pytest -m gpu (all, no -x)"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06b_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r06b_pytest_gpu.log | tail -8 | cut -c1-300
