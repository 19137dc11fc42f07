#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2; do echo "== iCub IK control, per-env poll word"; timeout 300 python tools/bench_icub.py --envs 32768 --steps 40 2>&1 | grep -vE "amdgpu.ids" | tail -1 | cut -c1-300; done
for E in own job; do
PBRE_BENCH_HOST_JOB_ENGINE=$( [ $E = job ] && echo 1 || echo 0 ) timeout 600 python bench.py --no-cpu-baseline --no-shards $( [ $E = job ] && echo --no-other-configs ) 2> gpurun_out/r06v_bench.err | tail -1 > gpurun_out/r06v_bench_$E.json
python - <<PY
import json
d=json.load(open("gpurun_out/r06v_bench_$E.json"))
h=d["host_inclusive"]; print("$E engine: host", {k: h.get(k) for k in ("value","ms_per_step","host_phase_ms_per_call","engine","complex_envs_at_the_end","one_launch_steps","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
oc = d.get("other_configs") or {}; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
if oc.get("icub_reach"): print("icub steady", json.dumps(oc["icub_reach"].get("steady_random_actions"))[:600])
PY
done
timeout 900 python -m pytest tests/test_gpu_icub.py tests/test_gpu_contention.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | tail -2 | cut -c1-300
