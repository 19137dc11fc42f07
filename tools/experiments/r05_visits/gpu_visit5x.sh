#!/bin/bash
# Round 5, visit X: the iCub pipeline's complex envs on the lane-group mapping (kw_step_list, PBRE_ICUB_RC_ROWS=1, default) against kw_quad_rc (0):
# timelines under joint and Cartesian control at 32768 envs, then the iCub GPU tests
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for M in joint ik; do for V in 0 1; do
  A=""; [ $M = joint ] && A="--joint"
  rm -rf gpurun_out/prof_icubx
  (cd /tmp && PBRE_ICUB_RC_ROWS=$V timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icubx -o run -- python $ROOTDIR/tools/icub_steady.py --desync --steps 1200 $A > $ROOTDIR/gpurun_out/icubx.log 2>&1)
  t=$(find gpurun_out/prof_icubx -name "*kernel_trace.csv" | head -1)
  echo "=== $M control, PBRE_ICUB_RC_ROWS=$V"; tail -1 gpurun_out/icubx.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([(w['ms_per_step'], w['complex_envs']) for w in d['windows']])" 2>/dev/null
  [ -n "$t" ] && python tools/trace_icub_steps.py $t 200 1 | grep -E "^step|^dur kw_quad|^dur kw_step|^span|^gap" | cut -c1-220
  find gpurun_out/prof_icubx -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icubx -name "*.db" -delete
done; done | tee gpurun_out/r05x_rc_rows.txt
timeout 1200 python -m pytest tests/test_gpu_icub.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | tail -8
