#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel-trace stats.  Logs -> gpurun_out/
# usage: tools/gpu_round.sh <tag> [bench args...]
TAG=${1:-r01}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu_$TAG.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke_$TAG.log
echo "== bench"
timeout 900 python bench.py "$@" 2>&1 | tail -3 | tee gpurun_out/bench_$TAG.log
echo "== rocprof"
ROOTDIR=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-other-configs --no-host-path --preroll 200 > $ROOTDIR/gpurun_out/rocprof_$TAG.log 2>&1)
tail -3 gpurun_out/rocprof_$TAG.log
find gpurun_out/prof_$TAG -name "*.csv" | head
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
echo "== iCub / hands benches"
timeout 300 python tools/bench_icub.py --envs 32768 --steps 20 2>&1 | tail -1 | tee gpurun_out/icub_bench_$TAG.json
timeout 300 python tools/bench_icub.py --envs 32768 --steps 20 --joint 2>&1 | tail -1 | tee -a gpurun_out/icub_bench_$TAG.json
timeout 300 python tools/bench_hands.py --envs 8192 --steps 20 2>&1 | tail -1 | tee gpurun_out/hands_bench_$TAG.json
timeout 300 python tools/bench_hands.py --envs 8192 --steps 20 --joint 2>&1 | tail -1 | tee -a gpurun_out/hands_bench_$TAG.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_icub_$TAG -o run -- python $ROOTDIR/tools/bench_icub.py --envs 32768 --steps 20 > $ROOTDIR/gpurun_out/rocprof_icub_$TAG.log 2>&1)
f=$(find gpurun_out/prof_icub_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
find gpurun_out/prof_icub_$TAG -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out/prof_icub_$TAG -name "*.db" -delete
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_hands_$TAG -o run -- python $ROOTDIR/tools/bench_hands.py --envs 8192 --steps 20 > $ROOTDIR/gpurun_out/rocprof_hands_$TAG.log 2>&1)
f=$(find gpurun_out/prof_hands_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
find gpurun_out/prof_hands_$TAG -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out/prof_hands_$TAG -name "*.db" -delete
