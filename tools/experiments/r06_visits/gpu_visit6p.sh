#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in 0 1 2; do
  echo "== PBRE_ASYNC_D2H=$V (bench context)"
  PBRE_ASYNC_D2H=$V timeout 600 python bench.py --no-cpu-baseline --no-shards $( [ $V != 0 ] && echo --no-other-configs ) 2> gpurun_out/r06p_bench.err | tail -1 > gpurun_out/r06p_bench_$V.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r06p_bench_$V.json"))
h=d["host_inclusive"]; print("host", {k: h.get(k) for k in ("value","ms_per_step","ms_per_call_last_8","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
oc = d.get("other_configs") or {}; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
PY
done 2>&1 | tee gpurun_out/r06p_host_modes.txt
PROBE_TORCH=1 timeout 600 python tools/host_async_probe.py 2>&1 | grep -E "pipelined" | cut -c1-300 | tee -a gpurun_out/r06p_host_modes.txt
timeout 900 python -m pytest tests/test_gpu_icub.py tests/test_gpu_contention.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | tail -3 | cut -c1-300
