#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in 0 2 1; do
  echo "== PBRE_ASYNC_D2H=$V (bench context)"
  PBRE_ASYNC_D2H=$V timeout 600 python bench.py --no-cpu-baseline --no-shards --no-other-configs 2> gpurun_out/r06q_bench.err | tail -1 > gpurun_out/r06q_bench_$V.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r06q_bench_$V.json"))
h=d["host_inclusive"]; print("host", {k: h.get(k) for k in ("value","ms_per_step","ms_per_call_last_8","error")}, "sync", h.get("synchronous",{}).get("ms_per_step"))
PY
done 2>&1 | tee gpurun_out/r06q_host_modes.txt
for V in 0 2; do PBRE_ASYNC_D2H=$V PROBE_TORCH=0 timeout 600 python tools/host_async_probe.py 2>&1 | grep -E "step_pipelined|^pipelined:" | cut -c1-300; done | tee -a gpurun_out/r06q_host_modes.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pipelined_host or staged" 2>&1 | grep -vE "^/opt/amdgpu" | tail -3 | cut -c1-300
