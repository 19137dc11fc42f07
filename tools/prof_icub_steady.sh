#!/bin/bash
# Kernel durations of the iCub pipeline in the stationary mix (tools/icub_steady.py: push env, random actions, auto-reset): rocprofv3
# --kernel-trace, mean duration per kernel over the LAST THIRD of the launches.   usage: tools/prof_icub_steady.sh <tag> [icub_steady args]
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
rm -rf gpurun_out/prof_icubs_$TAG
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_icubs_$TAG -o run -- python $ROOTDIR/tools/icub_steady.py --desync "$@" > $ROOTDIR/gpurun_out/icubs_$TAG.log 2>&1)
tail -1 gpurun_out/icubs_$TAG.log | cut -c1-600
t=$(find gpurun_out/prof_icubs_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" gpurun_out/icubs_${TAG}_kernels.json <<'PY'
import csv, sys, collections, json, re
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.match(r"(?:void )?(?:pbre::)?(\w+)", r["Kernel_Name"])
    per[m.group(1) if m else r["Kernel_Name"][:30]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = {}
for k, v in per.items():
    v.sort(); tail = v[len(v) * 2 // 3:]
    out[k] = {"launches": len(v), "mean_us_last_third": sum(e - s for s, e in tail) / len(tail) / 1e3}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -kv[1]["mean_us_last_third"] * kv[1]["launches"])[:12]:
    print("%-28s launches %6d  mean (last third) %8.1f us" % (k, d["launches"], d["mean_us_last_third"]))
PY
find gpurun_out/prof_icubs_$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/prof_icubs_$TAG -name "*.db" -delete
