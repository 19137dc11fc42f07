#!/bin/bash
# One GPU-box visit of round 2: the GPU tests (all of them, no -x), the bench at N=1, A/B runs of build variants.
# Logs -> gpurun_out/<tag>_*        usage: tools/gpu_visit.sh <tag> [variant suffixes...]
TAG=${1:-r02b}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOTDIR=$(pwd)
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -70
echo "== bench N=1"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err; cut -c1-1800 gpurun_out/${TAG}_bench.json
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=d.get('fresh_reset') or {}
print('%s: steady %.1f M (%.4f ms)  fresh %.1f M (%.4f ms)  k_fast %.4f ms  complex/step %.1f' % (sys.argv[2], d['value']/1e6, d['ms_per_step'], f.get('value',0)/1e6, f.get('ms_per_step',0), d['roofline']['kernel_ms'], d['config'].get('complex_envs_per_step_timed_region_rank0',-1)))
" "$1" "$2"; }
for V in "" "$@"; do
  LIB=$ROOTDIR/pybullet-robot-envs_amd/csrc/libpbre${V:+_$V}.so
  for E in 131072 16384; do
    PBRE_LIB=$LIB timeout 600 python bench.py --envs $E --no-cpu-baseline --no-other-configs --no-host-path 2>/dev/null | tail -1 > gpurun_out/${TAG}_ab_${V:-default}_$E.json
    short gpurun_out/${TAG}_ab_${V:-default}_$E.json "lib=${V:-default} envs=$E"
  done
done
