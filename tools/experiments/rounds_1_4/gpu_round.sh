#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel-trace stats + PMC passes of the bench command, the iCub / hands
# side benches, the N=2 control flow on one device.  Logs -> gpurun_out/<tag>_*; tools/collect_profiles.sh copies the summaries
# into profiles/.        usage: tools/gpu_round.sh <tag>
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTDIR=$(pwd)
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sed -n 2,3p
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -6 | cut -c1-300
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/${TAG}_smoke.log
echo "== bench"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json
echo "== rocprofv3 --kernel-trace --stats (same command, CPU baseline leg off)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/compact_stats.py $f gpurun_out/${TAG}_kernel_stats.csv && head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,100-
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_steps.py $t 8 | tail -3
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
echo "== PMC HBM / SQ (separate passes, no tracing domains)"
bash tools/pmc.sh $TAG --steps 20 --warmup 3 2>&1 | grep -E "k_fast<7>|==" | head -6
python tools/pmc_json.py $TAG 131072 2>&1 | tail -9
bash tools/pmc_sq.sh $TAG --steps 20 --warmup 3 2>&1 | tail -30 | grep -E "valu_insts_per_wave|valu_active|wait_any" | head -8
echo "== Panda: stationary step time around the machine-filling batch size"
timeout 600 python tools/tail_probe.py 2>&1 | grep "^{" | tee gpurun_out/${TAG}_tail_probe.json | cut -c1-260
echo "== iCub / hands benches"
rm -f gpurun_out/${TAG}_icub_steady.json gpurun_out/${TAG}_icub_bench.json
# (bench_icub.py: 1500 zero-action steps before the timed ones, see its --warm; "--warm 0" = the burst right after reset())
for L in 1 0; do for N in 32768 131072; do for M in "" "--joint"; do
  PBRE_ICUB_LANE=$L timeout 300 python tools/bench_icub.py --envs $N --steps 20 $M 2>&1 | tail -1 | sed "s/^{/{\"PBRE_ICUB_LANE\": $L, /" | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260
done; done; done
timeout 300 python tools/bench_icub.py --envs 262144 --steps 20 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260
for L in 1 0; do PBRE_ICUB_LANE=$L timeout 300 python tools/bench_icub.py --envs 16384 --steps 20 2>&1 | tail -1 | sed "s/^{/{\"PBRE_ICUB_LANE\": $L, /" | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260; done
timeout 300 python tools/bench_icub.py --envs 32768 --steps 20 --warm 0 2>&1 | tail -1 | sed 's/^{/{"protocol": "burst right after reset()", /' | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260
timeout 300 python tools/icub_sustain.py --envs 32768 --windows 5 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260
PBRE_ICUB_LANE=0 timeout 300 python tools/icub_sustain.py --envs 32768 --windows 5 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_bench.json | cut -c1-260
echo "== iCub: stationary mix under random actions (auto-reset), lane-per-env pipeline and lane-group kernel; kernel trace of the pipeline"
for N in 16384 32768; do for M in "" "--joint"; do
  PBRE_ICUB_LANE=1 timeout 600 python tools/icub_steady.py --envs $N --steps 1000 --window 250 $M 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
  PBRE_ICUB_LANE=0 timeout 600 python tools/icub_steady.py --envs $N --steps 1000 --window 250 $M 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
done; done
timeout 600 python tools/icub_steady.py --envs 131072 --steps 1000 --window 250 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
timeout 600 python tools/icub_steady.py --envs 262144 --steps 750 --window 250 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
timeout 600 python tools/icub_steady.py --envs 262144 --steps 750 --window 250 --joint 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_steady.json | cut -c1-400
echo "== iCub push with a scripted policy that drives the hand at the object (robot-object contacts in 10-40 % of the envs)"
rm -f gpurun_out/${TAG}_icub_push_soak.json
for N in 8192 32768 131072; do
  PBRE_ICUB_LANE=1 timeout 900 python tools/icub_push_soak.py --envs $N --steps 900 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_push_soak.json | cut -c1-260
  PBRE_ICUB_LANE=0 timeout 900 python tools/icub_push_soak.py --envs $N --steps 900 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_icub_push_soak.json | cut -c1-260
done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_icub_$TAG -o run -- python $ROOTDIR/tools/icub_steady.py --envs 32768 --steps 750 --window 250 > $ROOTDIR/gpurun_out/${TAG}_icub_rocprof.log 2>&1)
t=$(find gpurun_out/prof_icub_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_tail.py $t --last 200 | tee gpurun_out/${TAG}_icub_kernel_trace_tail.txt
f=$(find gpurun_out/prof_icub_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/compact_stats.py $f gpurun_out/${TAG}_icub_kernel_stats.csv
find gpurun_out/prof_icub_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icub_$TAG -name "*.db" -delete
bash tools/pmc_icub.sh $TAG 2>&1 | grep -E "valu_per_wave|valu_over|wait_any_over" | head -6
timeout 300 python tools/bench_hands.py --envs 8192 --steps 20 2>&1 | tail -1 | tee gpurun_out/${TAG}_hands_bench.json | cut -c1-300
echo "== bench N=2 on one device (control flow, gloo-staged gather)"
PBRE_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --preroll 200 2> gpurun_out/${TAG}_bench2.err | tail -1 > gpurun_out/${TAG}_bench2.json; echo rc=$?; cut -c1-300 gpurun_out/${TAG}_bench2.json
