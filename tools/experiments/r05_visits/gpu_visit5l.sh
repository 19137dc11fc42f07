#!/bin/bash
# Round 5, visit L: env.step() as ONE launch (k_fused, PBRE_FUSED=1, default) against the two kernels on two streams (PBRE_FUSED=0), one
# library, one box: fresh / stationary step time at 16384 and 131072 envs, the step-kernel timeline, then the Panda GPU tests.
export TMPDIR=/tmp
for r in 1 2; do
for V in 1 0; do
  echo "--- PBRE_FUSED=$V"; PBRE_FUSED=$V timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
done; done | tee gpurun_out/r05l_fused_ab.txt
for N in 131072 16384; do echo "== $N envs" | tee -a gpurun_out/r05l_step_kernels.txt; bash tools/trace_panda_steady3.sh $N r05l_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span|\"value\"" | cut -c1-200 | tee -a gpurun_out/r05l_step_kernels.txt; done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -vE "^/opt/amdgpu" | tail -5
