#!/bin/bash
# Round 5, visit A: GPU parity tests (incl. the residual-threshold cases), smoke, the default bench line.   usage: tools/gpu_visit5a.sh <tag>
TAG=${1:-r05a}
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sed -n 2,3p
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -25 | cut -c1-400
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k: d[k] for k in ("value","ms_per_step","repeats","first_timed_region")})
print("fresh", d["fresh_reset"]["ms_per_step"], "rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:1500])
print("shards", json.dumps(d.get("shards"))[:1200])
PY
