#!/bin/bash
# Round 6, visit O: whole GPU suite on the build with the residual exit's closed form + one-launch RT step, pipelined host path; bench
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06o_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r06o_pytest_gpu.log | tail -8 | cut -c1-300
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/r06o_bench.err | tail -1 > gpurun_out/r06o_bench.json; tail -2 gpurun_out/r06o_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06o_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step"])
print("fresh", d["fresh_reset"]["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print("rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:900])
h=d["host_inclusive"]; print("host", {k: h.get(k) for k in ("value","ms_per_step","ms_per_call_last_8","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
print("shards", json.dumps({k: (v.get("fresh_ms_per_step"), v.get("stationary_ms_per_step")) for k, v in d["shards"].items() if k.isdigit()}), json.dumps(d["shards"].get("projection"))[:300])
oc = d["other_configs"]; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
PY
