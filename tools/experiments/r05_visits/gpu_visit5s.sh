#!/bin/bash
# Round 5, visit S: timeline of the iCub lane pipeline's kernels in the stationary mix (IK control and joint control): where is the step's time?
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for M in ik joint; do
  A=""; [ $M = joint ] && A="--joint"
  rm -rf gpurun_out/prof_icubt_$M
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icubt_$M -o run -- python $ROOTDIR/tools/icub_steady.py --desync --steps 1200 $A > $ROOTDIR/gpurun_out/icubt_$M.log 2>&1)
  tail -1 gpurun_out/icubt_$M.log | cut -c1-300
  t=$(find gpurun_out/prof_icubt_$M -name "*kernel_trace.csv" | head -1)
  echo "=== $M"; [ -n "$t" ] && python tools/trace_icub_steps.py $t 200 2 | cut -c1-400
  find gpurun_out/prof_icubt_$M -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icubt_$M -name "*.db" -delete
done | tee gpurun_out/r05s_icub_timeline.txt
