#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_vec_env.py -q -x -m gpu > gpurun_out/r04o_pytest.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04o_pytest.log | tail -4 | cut -c1-300
for N in 131072 16384; do
bash tools/trace_panda_steady3.sh $N r04o_$N 2>&1 | tail -6
done
timeout 600 python tools/tail_probe.py --sizes 4096,16384,32768,65536,131072 --preroll 1100 --steps 600 2>&1 | grep "^{" | tee gpurun_out/r04o_shards.json | cut -c1-330
