#!/bin/bash
# Round 5, visit U: timeline of the iCub pipeline (IK control, 32768 envs) with PBRE_IK_OVERLAP=0 and =1 on one box, twice each
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for r in 1 2; do for V in 0 1; do
  rm -rf gpurun_out/prof_icubu
  (cd /tmp && PBRE_IK_OVERLAP=$V timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icubu -o run -- python $ROOTDIR/tools/icub_steady.py --desync --steps 1200 > $ROOTDIR/gpurun_out/icubu.log 2>&1)
  t=$(find gpurun_out/prof_icubu -name "*kernel_trace.csv" | head -1)
  echo "=== PBRE_IK_OVERLAP=$V (run $r)"; [ -n "$t" ] && python tools/trace_icub_steps.py $t 200 1 | grep -E "^step|^dur|^span|^gap|^idle|^start kw_quad|^end kw_lane" | cut -c1-200
  find gpurun_out/prof_icubu -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icubu -name "*.db" -delete
done; done | tee gpurun_out/r05u_icub_timeline_ab.txt
timeout 600 python -m pytest tests/test_gpu_icub.py -m gpu -q -x -k "overlap or pipeline" 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
