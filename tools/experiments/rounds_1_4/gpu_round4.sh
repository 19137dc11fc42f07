#!/bin/bash
# One GPU-box visit of round 4: parity tests, smoke, bench, rocprofv3 kernel-trace stats + PMC passes of the bench command, small shards,
# the iCub / hands side benches and counters, the N=2 control flow on one device.  Logs -> gpurun_out/<tag>_*;
# tools/collect_profiles4.sh copies the summaries into profiles/.        usage: tools/gpu_round4.sh <tag>
TAG=${1:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTDIR=$(pwd)
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sed -n 2,3p
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -4 | cut -c1-300
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json
echo "== rocprofv3 --kernel-trace --stats (same command, CPU baseline leg off)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-shards > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/compact_stats.py $f gpurun_out/${TAG}_kernel_stats.csv && head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,100-
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_steps.py $t 8 | tail -3
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
echo "== PMC HBM / SQ of the stationary headline (separate passes, no tracing domains)"
bash tools/profile_r04.sh $TAG 131072 2>&1 | tail -12
echo "== the same counters for a 16384-env shard (k_fast_pair + k_row_list)"
bash tools/profile_r04.sh ${TAG}_16384 16384 2>&1 | tail -12
echo "== Panda: small shards (the per-GPU batches of the strong-scaling split), stationary + fresh"
rm -f gpurun_out/${TAG}_small_shards.json
for E in 16384 32768 65536 131072; do
  timeout 600 python bench.py --envs $E --no-cpu-baseline --no-other-configs --no-host-path 2> /dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d.get('fresh_reset') or {}
print(json.dumps({'envs': $E, 'stationary_ms_per_step': d['ms_per_step'], 'stationary_env_steps_per_s': d['value'], 'repeats_ms': d.get('repeats',{}).get('ms_per_step'), 'fresh_ms_per_step': f.get('ms_per_step'), 'fresh_env_steps_per_s': f.get('value'), 'k_fast_ms': d['roofline']['kernel_ms'], 'k_fast_variant': d.get('k_fast_variant'), 'complex_envs_per_step': d['config'].get('complex_envs_per_step_timed_region_rank0')}))" | tee -a gpurun_out/${TAG}_small_shards.json | cut -c1-300
done
echo "== Panda: shard sizes, 600-step stationary windows (tools/tail_probe.py)"
timeout 600 python tools/tail_probe.py --sizes 4096,8192,16384,32768,65536,131072 --preroll 1100 --steps 600 2>&1 | grep "^{" | tee gpurun_out/${TAG}_shard_sweep.json | cut -c1-260
echo "== Panda: kernel durations of the stationary step, distribution over 250 steps"
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N ${TAG}_trace_$N 2>&1 | grep -E "min |span" | tee -a gpurun_out/${TAG}_step_kernels.txt; done
echo "== reset time"
python tools/reset_time.py 2>&1 | grep "^{" | tee gpurun_out/${TAG}_reset_time.json
echo "== Panda: stationary step time around the machine-filling batch size"
timeout 600 python tools/tail_probe.py 2>&1 | grep "^{" | tee gpurun_out/${TAG}_tail_probe.json | cut -c1-260
echo "== iCub / hands benches"
rm -f gpurun_out/${TAG}_icub_steady.json gpurun_out/${TAG}_icub_bench.json
for L in 1 0; do for N in 32768 131072; do for M in "" "--joint"; do
  PBRE_ICUB_LANE=$L timeout 300 python tools/bench_icub.py --envs $N --steps 20 $M 2>&1 | tail -1 | sed "s/^{/{\"PBRE_ICUB_LANE\": $L, /" | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260
done; done; done
for N in 16384 32768; do for M in "" "--joint"; do
  PBRE_ICUB_LANE=1 timeout 600 python tools/icub_steady.py --desync --envs $N --steps 1500 --window 250 $M 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
done; done
PBRE_ICUB_LANE=0 timeout 600 python tools/icub_steady.py --desync --envs 32768 --steps 1500 --window 250 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
timeout 600 python tools/icub_steady.py --desync --envs 131072 --steps 1500 --window 250 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
bash tools/prof_icub_steady.sh $TAG 2>&1 | tail -9; cp gpurun_out/icubs_${TAG}_kernels.json gpurun_out/${TAG}_icub_steady_kernels.json
bash tools/pmc_icub.sh $TAG 2>&1 | grep -E "valu_per_wave|valu_over|wait_any_over" | head -6
bash tools/pmc_icub_hbm.sh $TAG 2>&1 | tail -6
timeout 300 python tools/bench_hands.py --envs 8192 --steps 20 2>&1 | tail -1 | tee gpurun_out/${TAG}_hands_bench.json | cut -c1-400
bash tools/pmc_hands.sh $TAG 2>&1 | grep -E "valu_insts_per_wave|valu_active|hbm_bytes" | head -4
echo "== bench N=2 on one device (control flow, gloo-staged gather)"
PBRE_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --preroll 200 2> gpurun_out/${TAG}_bench2.err | tail -1 > gpurun_out/${TAG}_bench2.json; echo rc=$?; cut -c1-300 gpurun_out/${TAG}_bench2.json
