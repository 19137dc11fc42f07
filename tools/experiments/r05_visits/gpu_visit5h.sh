#!/bin/bash
# Round 5, visit H: clamp-free motor stages in the waves with robot-object rows only -- default build against libpbre_before.so (slot-count
# specialisation only), same box; kernel durations of the stationary step.
export TMPDIR=/tmp
D=$(pwd)/pybullet-robot-envs_amd/csrc
for r in 1 2; do
for V in "" _before; do
  echo "--- libpbre$V"; PBRE_LIB=$D/libpbre$V.so timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-40,130-260
done; done | tee gpurun_out/r05h_chain_ab.txt
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N r05h_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | tee -a gpurun_out/r05h_step_kernels.txt; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
