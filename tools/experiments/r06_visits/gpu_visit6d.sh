#!/bin/bash
# Round 6, visit D: the whole GPU suite on the build with the hull narrow phase, the ADVICE r5 fixes (second version of the IK timeout poison)
# and the contention tests; a bench line with the region-level kernel_ms; bench.py --gpus 8 control flow on one device over tests/fake_rccl
export TMPDIR=/tmp
mkdir -p gpurun_out
ROOTDIR=$(pwd)
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06d_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r06d_pytest_gpu.log | tail -12 | cut -c1-300
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/r06d_bench.err | tail -1 > gpurun_out/r06d_bench.json; tail -2 gpurun_out/r06d_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06d_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step"])
print("fresh", d["fresh_reset"]["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "sampled", d["roofline"]["kernel_ms_sampled_pairs_mean"], "frac", d["roofline"]["frac"])
print("valu", {k: d["valu"][k] for k in ("achieved","frac","frac_measured_fma")})
print("host", json.dumps(d["host_inclusive"])[:300])
oc = d["other_configs"]; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
PY
echo "== bench N=8 on one device: context-owned exchanges over tests/fake_rccl (control flow of the driver's SCALE run)"
PBRE_BENCH_ONE_DEVICE=1 PBRE_BENCH_CTX_COMM=force FAKE_RCCL_DEVICE=1 PBRE_RCCL_LIB=$ROOTDIR/tests/fake_rccl/build/libfake_rccl.so timeout 900 python bench.py --gpus 8 --steps 20 --preroll 200 2> gpurun_out/r06d_bench8.err | tail -1 > gpurun_out/r06d_bench8.json; echo rc=$?; tail -3 gpurun_out/r06d_bench8.err | cut -c1-300
python - <<PY
import json
d=json.load(open("gpurun_out/r06d_bench8.json"))
print({k: d.get(k) for k in ("value","ms_per_step","n_gpus","closed_loop","sharded_consumers_no_gather")}, d["config"]["rccl"])
PY
