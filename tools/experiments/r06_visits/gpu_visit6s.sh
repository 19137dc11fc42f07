#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for Q in 8 16 2; do
GPU_MAX_HW_QUEUES=$Q PBRE_BENCH_NO_RT=1 timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2> gpurun_out/r06s_bench.err | tail -1 > gpurun_out/r06s_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06s_bench.json"))
h=d["host_inclusive"]; print("GPU_MAX_HW_QUEUES=$Q", {k: d[k] for k in ("value","ms_per_step")}, "host", {k: h.get(k) for k in ("ms_per_step","host_phase_ms_per_call","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
PY
done
PBRE_SIDE_DEBUG=1 PBRE_BENCH_NO_RT=1 timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2>&1 >/dev/null | grep "side-stream probe" | tail -12
