#!/bin/bash
# Round 5, visit T: the iCub pipeline with the IK targets handed over per env (PBRE_IK_OVERLAP=1, default) against the kernel-level
# dependency (0): stationary mix under Cartesian control at 32768 envs, the kernels' timeline, then the iCub GPU tests
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for r in 1 2; do for V in 1 0; do
  echo "--- PBRE_IK_OVERLAP=$V"; PBRE_IK_OVERLAP=$V timeout 300 python tools/icub_steady.py --desync --steps 1500 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print([(w['ms_per_step'], w['complex_envs']) for w in d['windows']])"
done; done | tee gpurun_out/r05t_ik_overlap_ab.txt
rm -rf gpurun_out/prof_icubt_ik
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icubt_ik -o run -- python $ROOTDIR/tools/icub_steady.py --desync --steps 1200 > $ROOTDIR/gpurun_out/icubt_ik.log 2>&1)
t=$(find gpurun_out/prof_icubt_ik -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python tools/trace_icub_steps.py $t 200 2 | cut -c1-300 | tee -a gpurun_out/r05t_ik_overlap_ab.txt
find gpurun_out/prof_icubt_ik -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icubt_ik -name "*.db" -delete
timeout 600 python -m pytest tests/test_gpu_icub.py -m gpu -q -x -k "overlap or pipeline" 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
