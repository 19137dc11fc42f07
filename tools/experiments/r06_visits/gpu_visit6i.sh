#!/bin/bash
# Round 6, visit I: why is the host path 4 x slower in tools/host_async_probe.py than in bench.py?  (with / without torch's pinned copies first)
export TMPDIR=/tmp
mkdir -p gpurun_out
for T in 1 0; do echo "== PROBE_TORCH=$T"; PROBE_TORCH=$T timeout 600 python tools/host_async_probe.py 2>&1 | grep -vE "amdgpu.ids" | cut -c1-400; done
echo "== RT: sweeps / step times with the closed form (default) and the explicit rows (PBRE_SEQ flags)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "residual" 2>&1 | grep -vE "^/opt/amdgpu" | tail -4 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-shards --no-host-path 2> gpurun_out/r06i_bench.err | tail -1 > gpurun_out/r06i_bench.json; tail -2 gpurun_out/r06i_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06i_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}); print("rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:900])
PY
