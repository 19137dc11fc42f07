#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2> gpurun_out/r06r_bench.err | tail -1 > gpurun_out/r06r_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06r_bench.json"))
h=d["host_inclusive"]; print("host", {k: h.get(k) for k in ("value","ms_per_step","ms_per_call_last_8","host_phase_ms_per_call","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
PY
PBRE_BENCH_NO_RT=1 timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2> gpurun_out/r06r_bench.err | tail -1 > gpurun_out/r06r_bench2.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06r_bench2.json"))
h=d["host_inclusive"]; print("host (no RT side key before it)", {k: h.get(k) for k in ("value","ms_per_step","host_phase_ms_per_call","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
PY
