#!/bin/bash
# Round 6, visit A: the VALU issue-rate micro-benchmark, the GPU suite on the build with the ADVICE r5 fixes, a bench line
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sed -n 2,3p
bash tools/ubench/valu_rate.sh
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06a_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r06a_pytest_gpu.log | tail -4 | cut -c1-300
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/r06a_bench.err | tail -1 > gpurun_out/r06a_bench.json; tail -2 gpurun_out/r06a_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06a_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step"])
print("fresh", d["fresh_reset"]["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print("rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:400])
print("host", json.dumps(d["host_inclusive"])[:300])
oc = d["other_configs"]; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
PY
