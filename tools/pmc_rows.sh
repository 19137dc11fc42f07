#!/bin/bash
# SQ counters of k_row_list over the last launches of a stationary run at one batch size (tools/tail_probe.py): VALU instructions per
# launch and per working wave (complex envs / 4), busy cycles.   usage: tools/pmc_rows.sh <envs> [tag]
N=${1:-65536}; TAG=${2:-rows}
ROOTDIR=$(pwd); export TMPDIR=/tmp
rm -rf gpurun_out/pmcrows_$TAG
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
(cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $ROOTDIR/gpurun_out/pmcrows_$TAG -o run -- python $ROOTDIR/tools/tail_probe.py --sizes $N > $ROOTDIR/gpurun_out/pmcrows_$TAG.log 2>&1)
grep "^{" gpurun_out/pmcrows_$TAG.log | cut -c1-260
f=$(find gpurun_out/pmcrows_$TAG -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" "$TAG" <<'PY'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    if "k_row_list<7>" in r["Kernel_Name"]:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)[-50:]
out = {c: sum(by[i].get(c, 0.0) for i in ids) / len(ids) for c in by[ids[-1]]}
out["launches_averaged"] = len(ids)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_rows_%s.json" % sys.argv[2], "w"), indent=1)
PY
find gpurun_out/pmcrows_$TAG -name "*.csv" -size +6M -delete; find gpurun_out/pmcrows_$TAG -name "*.db" -delete
