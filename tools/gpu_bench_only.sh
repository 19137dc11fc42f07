#!/bin/bash
# Re-run of the bench alone (run of record for profiles/<tag>_bench.json once the PMC summaries of the same round are committed: the
# bench line quotes them as roofline.traffic).   usage: tools/gpu_bench_only.sh <tag>
TAG=${1:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench.json
