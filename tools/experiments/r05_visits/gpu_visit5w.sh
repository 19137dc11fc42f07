#!/bin/bash
# Round 5, visit W: envs per wave of kw_quad_rc (PBRE_QUAD_RC_EPW = 16 (rounds 2-4), 4, 1): the iCub pipeline's timeline under Cartesian and joint
# control, 32768 envs, stationary mix; then the iCub GPU tests
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for M in ik joint; do for E in 16 4 1; do
  A=""; [ $M = joint ] && A="--joint"
  rm -rf gpurun_out/prof_icubw
  (cd /tmp && PBRE_QUAD_RC_EPW=$E timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icubw -o run -- python $ROOTDIR/tools/icub_steady.py --desync --steps 1200 $A > $ROOTDIR/gpurun_out/icubw.log 2>&1)
  t=$(find gpurun_out/prof_icubw -name "*kernel_trace.csv" | head -1)
  echo "=== $M control, PBRE_QUAD_RC_EPW=$E"; [ -n "$t" ] && python tools/trace_icub_steps.py $t 200 1 | grep -E "^step|^dur kw_quad|^span|^gap" | cut -c1-200
  find gpurun_out/prof_icubw -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icubw -name "*.db" -delete
done; done | tee gpurun_out/r05w_quad_rc_epw.txt
timeout 1200 python -m pytest tests/test_gpu_icub.py -m gpu -q -x 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
