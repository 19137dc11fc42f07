#!/bin/bash
# One GPU-box visit of round 2: measured parity numbers, the GPU tests (all of them, no -x), the bench at N=1 and the N=2
# control flow on one device.  Logs -> gpurun_out/<tag>_*
TAG=${1:-r02a}
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -2
echo "== parity report (hip)"
timeout 300 python tools/parity_report.py --lib hip --n 64 > gpurun_out/${TAG}_parity_report.json 2> gpurun_out/${TAG}_parity_report.err; tail -3 gpurun_out/${TAG}_parity_report.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_parity_report.json"))
for k,v in d.items():
    print(k, {a:v[a] for a in ('envs','steps','skipped_ambiguous','done_flips','compared')})
    print('   ', ' '.join('%s=%.2g'%(a,b) for a,b in v['worst'].items()))
PY
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | tail -60 | tee gpurun_out/${TAG}_pytest_gpu.log
echo "== bench N=1"
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -5 gpurun_out/${TAG}_bench.err; cut -c1-1500 gpurun_out/${TAG}_bench.json
echo "== bench N=2 on one device (control flow, gloo-staged gather)"
PBRE_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --preroll 200 2> gpurun_out/${TAG}_bench2.err | tail -1 > gpurun_out/${TAG}_bench2.json; echo rc=$?; tail -5 gpurun_out/${TAG}_bench2.err; cut -c1-1200 gpurun_out/${TAG}_bench2.json
