#!/bin/bash
# Round 5, visit V: after the row waves' fence went from system scope to workgroup scope: step time and k_fused durations (compare r05_fused_step_ab.txt:
# 131072 envs 0.1532 ms / k_fused median 149 us, 16384 envs 0.120 ms / 121 us), then the Panda GPU tests
export TMPDIR=/tmp
for r in 1 2; do timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260; done | tee gpurun_out/r05v_fence_scope.txt
for N in 131072 16384; do echo "== $N envs" | tee -a gpurun_out/r05v_fence_scope.txt; bash tools/trace_panda_steady3.sh $N r05v_trace_$N PBRE_BENCH_NO_RT=1 2>&1 | grep -E "min |span" | cut -c1-200 | tee -a gpurun_out/r05v_fence_scope.txt; done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
