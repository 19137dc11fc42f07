#!/bin/bash
# Round 5, visit B: all GPU tests, smoke, the default bench line, rocprofv3 kernel stats of the same command, PMC passes (HBM, SQ), the
# stationary step's kernel durations, bench N=2 on one device with the context-owned exchanges over tests/fake_rccl.   usage: tools/gpu_visit5b.sh <tag>
TAG=${1:-r05}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTDIR=$(pwd)
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -30 | cut -c1-300
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k: d[k] for k in ("value","ms_per_step","repeats","first_timed_region")})
print("fresh", d["fresh_reset"]["ms_per_step"], "rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:900])
print("roofline", json.dumps(d["roofline"])[:600])
PY
echo "== rocprofv3 --kernel-trace --stats (same command, CPU baseline leg off)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-shards > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/compact_stats.py $f gpurun_out/${TAG}_kernel_stats.csv && head -10 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,100-
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
echo "== PMC HBM / SQ of the stationary headline (separate passes, no tracing domains)"
bash tools/profile_r05.sh $TAG 131072 2>&1 | tail -14
echo "== kernel durations of the stationary step, distribution over 250 steps"
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N ${TAG}_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | tee -a gpurun_out/${TAG}_step_kernels.txt; done
echo "== bench N=2 on one device: context-owned exchanges over tests/fake_rccl (control flow of the N > 1 path incl. the closed loop)"
PBRE_BENCH_ONE_DEVICE=1 PBRE_BENCH_CTX_COMM=force FAKE_RCCL_DEVICE=1 PBRE_RCCL_LIB=$ROOTDIR/tests/fake_rccl/build/libfake_rccl.so timeout 600 python bench.py --gpus 2 --steps 20 --preroll 200 2> gpurun_out/${TAG}_bench2.err | tail -1 > gpurun_out/${TAG}_bench2.json; echo rc=$?; tail -3 gpurun_out/${TAG}_bench2.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench2.json"))
print({k: d.get(k) for k in ("value","ms_per_step","closed_loop","sharded_consumers_no_gather")}, d["config"]["rccl"])
PY
