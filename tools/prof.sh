#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/prof_$1/
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-other-configs --no-host-path --preroll 200 "$@" > $ROOTDIR/gpurun_out/rocprof_$TAG.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-60,150-
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
