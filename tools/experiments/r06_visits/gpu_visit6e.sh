#!/bin/bash
# Round 6, visit E: is the iCub hand-over's mismatch under contention the hand-over (PBRE_IK_OVERLAP=0 passes?); the pipelined host path
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in 1 0; do echo "== PBRE_IK_OVERLAP=$V"; PBRE_IK_OVERLAP=$V timeout 600 python -m pytest tests/test_gpu_contention.py -m gpu -q -k icub 2>&1 | grep -vE "^/opt/amdgpu" | grep -E "rows differ|passed|failed" | cut -c1-300; done
echo "== pipelined host path"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pipelined_host or staged_copies or hull" 2>&1 | grep -vE "^/opt/amdgpu" | tail -5 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-shards 2> gpurun_out/r06e_bench.err | tail -1 > gpurun_out/r06e_bench.json; tail -2 gpurun_out/r06e_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06e_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}); print("host", json.dumps(d["host_inclusive"])[:900])
PY
