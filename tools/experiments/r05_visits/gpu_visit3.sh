#!/bin/bash
# One GPU-box visit of round 3: selected GPU tests, then bench lines of the Panda headline at several shard sizes.
#   usage: tools/gpu_visit3.sh <tag> "<pytest selection>" [env sizes...]
TAG=${1:-r03a}; SEL=${2:-tests/test_gpu_parity.py}; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest $SEL"
timeout 1500 python -m pytest $SEL -m gpu -q -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/${TAG}_pytest_gpu.log | tail -40
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=d.get('fresh_reset') or {}
print('%s: steady %.1f M (%.4f ms)  fresh %.1f M (%.4f ms)  k_fast %.4f ms  complex/step %.1f' % (sys.argv[2], d['value']/1e6, d['ms_per_step'], f.get('value',0)/1e6, f.get('ms_per_step',0), d['roofline']['kernel_ms'], d['config'].get('complex_envs_per_step_timed_region_rank0',-1)))
" "$1" "$2"; }
for E in "$@"; do
  timeout 600 python bench.py --envs $E --no-cpu-baseline --no-other-configs --no-host-path 2> gpurun_out/${TAG}_bench_$E.err | tail -1 > gpurun_out/${TAG}_bench_$E.json
  short gpurun_out/${TAG}_bench_$E.json "envs=$E" || tail -5 gpurun_out/${TAG}_bench_$E.err
done
