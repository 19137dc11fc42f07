#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
echo "== RT tests, default lib"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "residual" 2>&1 | grep -vE "^/opt/amdgpu" | tail -5 | cut -c1-300
echo "== RT tests, PBRE_RT_CLOSED=0 variant of the env.step() TU"; PBRE_LIB=$C/libpbre_rtexp.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "residual" 2>&1 | grep -vE "^/opt/amdgpu" | tail -5 | cut -c1-300
echo "== bench (host path, RT side key)"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-shards 2> gpurun_out/r06l_bench.err | tail -1 > gpurun_out/r06l_bench.json; tail -2 gpurun_out/r06l_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06l_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}); h=d["host_inclusive"]; print("host pipelined", h.get("ms_per_step"), h.get("value"), "sync", h.get("synchronous",{}).get("ms_per_step"), h.get("error"))
print("rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:700])
PY
