#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/reset_time.py 2>&1 | grep "^{" | tee gpurun_out/r04q_reset_time.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs 2> gpurun_out/r04q_bench.err | tail -1 > gpurun_out/r04q_bench.json; tail -3 gpurun_out/r04q_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04q_bench.json').read())
print(d['value']/1e6, d['ms_per_step'], d['repeats']['ms_per_step'], d['fresh_reset']['ms_per_step'])
print(json.dumps(d['shards'])[:1500])
print(d['k_fast_variant'], d['nan_inf_guard'], d['roofline']['kernel_ms'])
PY
