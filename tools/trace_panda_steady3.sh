#!/bin/bash
# Kernel trace of the Panda's stationary step (bench.py's protocol, 1000-step pre-roll): durations of k_fast / k_row_list over the last
# launches and their start offsets.   usage: tools/trace_panda_steady3.sh <envs> <tag> [env assignments...]
N=${1:-131072}; TAG=${2:-tp3}; shift 2
export TMPDIR=/tmp
ROOTDIR=$(pwd)
rm -rf gpurun_out/prof_$TAG
(cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --envs $N --no-cpu-baseline --no-other-configs --no-host-path --no-fresh --no-shards > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
s=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && python tools/compact_stats.py $s gpurun_out/${TAG}_kernel_stats.csv && head -8 gpurun_out/${TAG}_kernel_stats.csv
[ -n "$t" ] && python tools/trace_steps.py $t 250 | tail -6
tail -1 gpurun_out/${TAG}_rocprof.log | cut -c1-200
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
