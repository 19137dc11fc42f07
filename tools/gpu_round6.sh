#!/bin/bash
# The GPU-box visit of round 6 that produces profiles/r05_*: parity tests, smoke, bench, rocprofv3 kernel-trace stats + PMC passes of the
# bench command, kernel durations of the stationary step, iCub / hands counters, bench N=8 on one device over tests/fake_rccl.
# Logs -> gpurun_out/<tag>_*; tools/collect_profiles5.sh copies the summaries into profiles/.        usage: tools/gpu_round6.sh <tag>
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTDIR=$(pwd)
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sed -n 2,3p
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -4 | cut -c1-300
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/${TAG}_smoke.log
echo "== PMC HBM / SQ of the stationary headline (separate passes, no tracing domains)"
bash tools/profile_r05.sh $TAG 131072 2>&1 | grep -E "k_fast|k_fused|k_row|valu_insts_per_wave|hbm_bytes_per_env_step|launches|valu_active" | head -40
echo "== the same counters for a 16384-env shard (k_fast_pair + k_row_list)"
bash tools/profile_r05.sh ${TAG}_16384 16384 0 2>&1 | grep -E "k_fast|k_fused|valu_insts_per_wave|hbm_bytes_per_env_step" | head -16
# (bench.py's `roofline.traffic` / `valu.sq_counters` are read from profiles/<tag>_pmc_*.json: this visit's counter summaries, copied here before the bench runs)
cp gpurun_out/${TAG}_pmc_hbm.json profiles/${TAG}_pmc_hbm.json 2>/dev/null; cp gpurun_out/${TAG}_pmc_sq.json profiles/${TAG}_pmc_sq.json 2>/dev/null
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step"], "first", d["first_timed_region"])
print("fresh", d["fresh_reset"]["ms_per_step"], d["fresh_reset"]["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("rt", json.dumps(d.get("solver_residual_threshold_1e-7"))[:600])
print("shards", json.dumps({k: (v.get("fresh_ms_per_step"), v.get("stationary_ms_per_step")) for k, v in d["shards"].items() if k.isdigit()}), json.dumps(d["shards"].get("projection"))[:300])
print("host", json.dumps(d["host_inclusive"])[:300]); print("cpu", json.dumps(d["cpu_baseline"])[:400])
oc = d["other_configs"]; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
PY
echo "== rocprofv3 --kernel-trace --stats (same command, CPU baseline leg off)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-shards > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/compact_stats.py $f gpurun_out/${TAG}_kernel_stats.csv && head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,100-
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
echo "== kernel durations of the stationary step, distribution over 250 steps"
rm -f gpurun_out/${TAG}_step_kernels.txt
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N ${TAG}_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | tee -a gpurun_out/${TAG}_step_kernels.txt; done
echo "== iCub / hands counters"
bash tools/pmc_icub.sh $TAG 2>&1 | grep -E "valu_per_wave|valu_over|wait_any_over" | head -6
bash tools/pmc_icub_hbm.sh $TAG 2>&1 | tail -6
bash tools/pmc_hands.sh $TAG 2>&1 | grep -E "valu_insts_per_wave|valu_active|hbm_bytes" | head -4
echo "== bench N=8 on one device: context-owned exchanges over tests/fake_rccl (control flow of the N > 1 path incl. the closed loop)"
PBRE_BENCH_ONE_DEVICE=1 PBRE_BENCH_CTX_COMM=force FAKE_RCCL_DEVICE=1 PBRE_RCCL_LIB=$ROOTDIR/tests/fake_rccl/build/libfake_rccl.so timeout 600 python bench.py --gpus 8 --steps 20 --preroll 200 2> gpurun_out/${TAG}_bench8.err | tail -1 > gpurun_out/${TAG}_bench8.json; echo rc=$?; tail -3 gpurun_out/${TAG}_bench8.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench8.json"))
print({k: d.get(k) for k in ("value","ms_per_step","closed_loop","sharded_consumers_no_gather")}, d["config"]["rccl"])
PY
