#!/bin/bash
# Round 5, visit D: the complex envs' chain -- clamp-free motor stages + slot-count specialisation against the round-4 chain (same box,
# libpbre_r4chain.so = -DPBRE_FREE_MOTOR_STAGES=0 -DPBRE_NRO_SPECIAL=0), kernel durations of the stationary step, Panda GPU tests.
TAG=${1:-r05d}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest (Panda parity)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -8 | cut -c1-300
echo "== A/B stationary step, default build vs round-4 chain"
bash tools/gpu_ab_lib.sh r4chain 16384,131072 2 2>&1 | tee gpurun_out/${TAG}_chain_ab.txt
echo "== kernel durations of the stationary step (default build)"
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N ${TAG}_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | tee -a gpurun_out/${TAG}_step_kernels.txt; done
echo "== the same with the round-4 chain"
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N ${TAG}_trace_r4_$N PBRE_BENCH_NO_RT=1 PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_r4chain.so 2>&1 | grep -E "min |span" | tee -a gpurun_out/${TAG}_step_kernels_r4chain.txt; done
