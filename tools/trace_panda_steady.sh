#!/bin/bash
# Kernel trace of the Panda's stationary step at one batch size: durations of k_fast / k_row_list over the last timed steps and their
# start offsets (tools/trace_tail.py, tools/trace_steps.py).   usage: tools/trace_panda_steady.sh <envs> [tag]
N=${1:-65536}; TAG=${2:-tp}
export TMPDIR=/tmp
ROOTDIR=$(pwd)
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/tools/tail_probe.py --sizes $N > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_tail.py $t --last 50 --marker k_fast && python tools/trace_steps.py $t 6 | tail -8
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
