#!/bin/bash
# Round 6, visit F: the stamped IK hand-over under contention (5 runs each test), the pipelined host path: DMA engine against the copy kernel
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_gpu_contention.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | grep -E "rows differ|passed|failed" | cut -c1-300; done
timeout 900 python -m pytest tests/test_gpu_icub.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | tail -3 | cut -c1-300
for V in "PBRE_ASYNC_D2H=0" "PBRE_ASYNC_D2H=1 PBRE_ASYNC_BLOCKS=32" "PBRE_ASYNC_D2H=1 PBRE_ASYNC_BLOCKS=128" "PBRE_ASYNC_D2H=1 PBRE_ASYNC_BLOCKS=512" "PBRE_ASYNC_D2H=0 HSA_ENABLE_SDMA=0"; do
  echo "== $V"
  env $V timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-shards 2> gpurun_out/r06f_bench.err | tail -1 > gpurun_out/r06f_bench.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r06f_bench.json"))
h=d["host_inclusive"]; print({k: d[k] for k in ("value","ms_per_step")}, "host pipelined", h.get("ms_per_step"), h.get("d2h_GBps_if_download_bound"), "sync", h.get("synchronous",{}).get("ms_per_step"), h.get("error"))
PY
done 2>&1 | tee gpurun_out/r06f_host_path_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pipelined_host" 2>&1 | grep -vE "^/opt/amdgpu" | tail -2 | cut -c1-300
