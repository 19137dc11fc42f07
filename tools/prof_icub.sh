#!/bin/bash
# rocprofv3 kernel-trace stats of the iCub bench (tools/bench_icub.py) -> gpurun_out/prof_icub_$1/ ; usage: prof_icub.sh TAG [bench args]
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_icub_$TAG -o run -- python $ROOTDIR/tools/bench_icub.py "$@" > $ROOTDIR/gpurun_out/rocprof_icub_$TAG.log 2>&1)
f=$(find gpurun_out/prof_icub_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-60s calls %6s  avg %9.1f us  total %6.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
find gpurun_out/prof_icub_$TAG -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out/prof_icub_$TAG -name "*.db" -delete
