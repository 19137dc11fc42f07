#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in 1 0; do for r in 1 2; do echo "== PBRE_IK_OVERLAP=$V"; PBRE_IK_OVERLAP=$V timeout 300 python tools/bench_icub.py --envs 32768 --steps 40 2>&1 | grep -vE "amdgpu.ids" | tail -1 | cut -c1-300; done; done
echo "== joint control"; timeout 300 python tools/bench_icub.py --envs 32768 --steps 40 --joint 2>&1 | grep -vE "amdgpu.ids" | tail -1 | cut -c1-300
PBRE_ICUB_TRACE=1700 timeout 300 python tools/bench_icub.py --envs 32768 --steps 40 2>&1 | grep "iCub step" | head -3 | cut -c1-300
PBRE_BENCH_NO_RT=1 timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2> gpurun_out/r06u_bench.err | tail -1 > gpurun_out/r06u_bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06u_bench.json"))
h=d["host_inclusive"]; print("host", {k: h.get(k) for k in ("value","ms_per_step","host_phase_ms_per_call","engine","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
PY
