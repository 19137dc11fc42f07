#!/bin/bash
# Round 5, visit K: where the stationary step's time goes between the kernels -- step-kernel timeline with the 168-VGPR k_fast build
# (PBRE_FAST3=2) and with the spill-free build only (PBRE_FAST3=0): kernel durations, idle gaps between steps, start lag of the second kernel.
export TMPDIR=/tmp
for V in 2 0; do
  echo "=== PBRE_FAST3=$V" | tee -a gpurun_out/r05k_step_kernels.txt
  for N in 131072; do bash tools/trace_panda_steady3.sh $N r05k_trace_${N}_f$V PBRE_BENCH_NO_RT=1 PBRE_FAST3=$V 2>&1 | grep -E "min |span|\"value\"" | cut -c1-220 | tee -a gpurun_out/r05k_step_kernels.txt; done
done
