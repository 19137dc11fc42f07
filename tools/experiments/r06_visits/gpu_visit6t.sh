#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
PBRE_BENCH_HOST_FRESH=1 PBRE_BENCH_NO_RT=1 timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2> gpurun_out/r06t_bench.err | tail -1 > gpurun_out/r06t_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06t_bench.json"))
h=d["host_inclusive"]; print("fresh engine in bench's process:", "host", {k: h.get(k) for k in ("ms_per_step","host_phase_ms_per_call","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
PY
