#!/bin/bash
# Round 5, visit P: reset() with the settle steps as one launch each (PBRE_FUSED=1, default) against the two-kernel settle step (0); then the
# Panda GPU tests
export TMPDIR=/tmp
for V in 1 0; do echo "--- PBRE_FUSED=$V"; PBRE_FUSED=$V timeout 300 python tools/reset_time.py 2>&1 | grep envs; done | tee gpurun_out/r05_reset_time.json
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -vE "^/opt/amdgpu" | tail -4
