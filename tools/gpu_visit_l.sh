#!/bin/bash
ROOTDIR=$(pwd); export TMPDIR=/tmp
for M in ik joint; do
  F=""; [ $M = joint ] && F="--joint"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icub_steady_$M -o run -- python $ROOTDIR/tools/icub_steady.py --envs 32768 --steps 500 --window 250 $F > $ROOTDIR/gpurun_out/rocprof_icub_steady_$M.log 2>&1)
  echo "== steady $M"; tail -1 gpurun_out/rocprof_icub_steady_$M.log | cut -c1-400
  f=$(find gpurun_out/prof_icub_steady_$M -name "*kernel_trace.csv" | head -1); python tools/trace_tail.py $f --last 100
  find gpurun_out/prof_icub_steady_$M -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icub_steady_$M -name "*.db" -delete
done
