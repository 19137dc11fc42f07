#!/bin/bash
# Round 5, visit C: A/B of the object block's closed form (PBRE_F_SEQ_OBJECT off / on through the env knob of bench: flags), the Panda GPU
# tests, SQ counters of the fresh and stationary k_fast.   usage: tools/gpu_visit5c.sh <tag>
TAG=${1:-r05c}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTDIR=$(pwd)
echo "== pytest (Panda parity + rccl)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-300
echo "== bench (closed form on)"
timeout 900 python bench.py --no-other-configs --no-cpu-baseline 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step"])
print("fresh", d["fresh_reset"]["ms_per_step"], d["fresh_reset"]["value"], "kernel_ms", d["roofline"]["kernel_ms"])
print("rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:700])
print("shards", json.dumps({k: (v.get("fresh_ms_per_step"), v.get("stationary_ms_per_step")) for k, v in d["shards"].items() if k.isdigit()}))
PY
echo "== bench (PBRE_SEQ_OBJECT=1: all object rows sequential, same box)"
PBRE_SEQ_OBJECT=1 timeout 900 python bench.py --no-other-configs --no-cpu-baseline --no-shards --no-host-path 2> /dev/null | tail -1 > gpurun_out/${TAG}_bench_seqobj.json
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_seqobj.json"))
print({k: d[k] for k in ("value","ms_per_step")}, "fresh", d["fresh_reset"]["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
echo "== SQ counters"
bash tools/profile_r05.sh $TAG 131072 2>&1 | grep -E "k_fast|valu_insts_per_wave|hbm_bytes_per_env_step|launches" | head -40
