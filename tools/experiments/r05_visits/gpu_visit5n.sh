#!/bin/bash
# Round 5, visit N: bench.py again with this round's counter summaries in profiles/ (the hands summary of the first pass was empty: kernel-name
# filter), and the rocprofv3 kernel-trace stats of the same command.
TAG=${1:-r05}
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 900 python bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print({k: d[k] for k in ("value","ms_per_step")}, d["repeats"]["ms_per_step"], "first", d["first_timed_region"])
print("fresh", d["fresh_reset"]["ms_per_step"], d["fresh_reset"]["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
oc = d["other_configs"]; print({k: (v.get("value"), v.get("ms_per_step")) if isinstance(v, dict) else v for k, v in oc.items()})
PY
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-shards > $ROOTDIR/gpurun_out/${TAG}_rocprof.log 2>&1)
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/compact_stats.py $f gpurun_out/${TAG}_kernel_stats.csv && head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,100-
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
