#!/bin/bash
# default step: does the simple waves' object-block fallback (a lane that fails obj_closed's bound -> its wave runs the 128 explicit sweeps) shape
# the UPPER TAIL of k_fused's duration?  kernel-duration distribution of 250 stationary steps, base build against -DPBRE_OC_NO_FALLBACK=1
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
: > gpurun_out/r06zd_tail_ab.txt
for rep in 1 2; do
for L in libpbre_nofb.so libpbre_ocnofb.so; do
  echo "== $L (131072 envs)" | tee -a gpurun_out/r06zd_tail_ab.txt
  bash tools/trace_panda_steady3.sh 131072 r06zd PBRE_BENCH_NO_RT=1 PBRE_LIB=$C/$L 2>&1 | grep -E "min |span" | tee -a gpurun_out/r06zd_tail_ab.txt
done
done
