#!/bin/bash
# Round 5, visit J: the stationary step's k_fast as two half-grid launches of the spill-free build (PBRE_FAST3=3, default) against the
# 168-VGPR build (PBRE_FAST3=2, round 4) and against neither (0), one library, one box; kernel durations of the stationary step.
export TMPDIR=/tmp
for r in 1 2; do
for V in 3 2 0; do
  echo "--- PBRE_FAST3=$V"; PBRE_FAST3=$V timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-40,130-260
done; done | tee gpurun_out/r05j_fast3_ab.txt
for N in 131072 16384; do bash tools/trace_panda_steady3.sh $N r05j_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | tee -a gpurun_out/r05j_step_kernels.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
