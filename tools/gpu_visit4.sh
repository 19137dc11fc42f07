#!/bin/bash
TAG=${1:-r02d}
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOTDIR=$(pwd)
echo "== pytest -m gpu (parity files only)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -12 | cut -c1-300
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=d.get('fresh_reset') or {}
print('%s: steady %.1f M (%.4f ms)  fresh %.1f M (%.4f ms)  ratio %.3f  k_fast %.4f ms  complex/step %.1f' % (sys.argv[2], d['value']/1e6, d['ms_per_step'], f.get('value',0)/1e6, f.get('ms_per_step',0), d['value']/max(f.get('value',1),1), d['roofline']['kernel_ms'], d['config'].get('complex_envs_per_step_timed_region_rank0',-1)))
" "$1" "$2"; }
for rep in 1 2; do
for V in "" base; do
  LIB=$ROOTDIR/pybullet-robot-envs_amd/csrc/libpbre${V:+_$V}.so
  PBRE_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-host-path 2>/dev/null | tail -1 > gpurun_out/${TAG}_ab_${V:-default}_$rep.json
  short gpurun_out/${TAG}_ab_${V:-default}_$rep.json "lib=${V:-default} rep=$rep"
done; done
echo "== PMC HBM (default lib)"
bash tools/pmc.sh $TAG --steps 20 --warmup 3 2>&1 | grep -E "k_fast|==" | head -12
python tools/pmc_json.py $TAG 131072 2>&1 | tail -12
