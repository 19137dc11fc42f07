#!/bin/bash
# SQ issue counters of the step kernels in one --pmc pass (8 SQ slots); no other tracing domains.
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
(cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $ROOTDIR/gpurun_out/pmcsq_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-other-configs --no-host-path --preroll 200 "$@" > $ROOTDIR/gpurun_out/pmcsq_$TAG.log 2>&1)
f=$(find gpurun_out/pmcsq_$TAG -name "*counter_collection.csv" | head -1)
echo "== $f"
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "k_fast<7>" in k or "k_fast_rc<7>" in k or "k_row_list<7>" in k:
        name = "k_fast<7>" if "k_fast<7>" in k else ("k_fast_rc<7>" if "k_fast_rc<7>" in k else "k_row_list<7>")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
for k, d in out.items():
    if d.get("SQ_WAVE_CYCLES"):
        d["valu_active_over_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_WAVE_CYCLES"]
        d["wait_any_over_wave_cycles"] = d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]
        d["wait_inst_any_over_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0) / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_WAVES"):
        d["valu_insts_per_wave"] = d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"]
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmcsq_%s.json" % sys.argv[1].split("pmcsq_")[1].split("/")[0], "w"), indent=1)
PY
find gpurun_out/pmcsq_$TAG -name "*.csv" -size +6M -delete; find gpurun_out/pmcsq_$TAG -name "*.db" -delete
