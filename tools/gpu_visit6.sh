#!/bin/bash
TAG=${1:-r02f}
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOTDIR=$(pwd)
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -15 | cut -c1-300
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=d.get('fresh_reset') or {}
print('%s: steady %.1f M (%.4f ms)  fresh %.1f M (%.4f ms)  ratio %.3f  k_fast %.4f ms  complex/step %.1f' % (sys.argv[2], d['value']/1e6, d['ms_per_step'], f.get('value',0)/1e6, f.get('ms_per_step',0), d['value']/max(f.get('value',1),1), d['roofline']['kernel_ms'], d['config'].get('complex_envs_per_step_timed_region_rank0',-1)))
" "$1" "$2"; }
run() { NAME=$1; V=$2; shift; shift
  LIB=$ROOTDIR/pybullet-robot-envs_amd/csrc/libpbre${V:+_$V}.so
  env PBRE_LIB=$LIB "$@" timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-host-path 2>/dev/null | tail -1 > gpurun_out/${TAG}_ab_$NAME.json
  short gpurun_out/${TAG}_ab_$NAME.json "$NAME"
}
for rep in 1 2; do
  run fused_$rep "" A=1
  run unfused_samebin_$rep "" PBRE_FUSE_ROWS=0
  run prefusion_$rep prio0 A=1
done
