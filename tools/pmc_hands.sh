#!/bin/bash
# Counters of the iCub-with-hands step kernel (kw_step<Shape128, DevLanes128, 14>): one SQ pass (issue / LDS), one HBM pass each for
# FETCH_SIZE and WRITE_SIZE; --pmc only, no other tracing domains.  Summary -> gpurun_out/pmc_hands_<tag>.json
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
run() { # name, counters...
    local name=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $ROOTDIR/gpurun_out/pmch_${TAG}_$name -o run -- python $ROOTDIR/tools/bench_hands.py --envs 8192 --steps 10 > $ROOTDIR/gpurun_out/pmch_${TAG}_$name.log 2>&1)
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python - $TAG <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {"source": "rocprofv3 --pmc (3 separate passes, tools/pmc_hands.sh), tools/bench_hands.py --envs 8192 --steps 10, 1 MI355X",
       "kernel": "kw_step<Shape128, DevLanes128, 14> (IK targets + task + observation)"}
for name in ("sq", "fetch", "write"):
    fs = glob.glob("gpurun_out/pmch_%s_%s/**/*counter_collection.csv" % (tag, name), recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "kw_step" in r["Kernel_Name"] and ("14>" in r["Kernel_Name"].split("(")[0] or "14, false>" in r["Kernel_Name"].split("(")[0]):      # (round 5: trailing RT flag)
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        out[c] = sum(v) / len(v)
if out.get("SQ_WAVES"):
    out["valu_insts_per_wave"] = out.get("SQ_INSTS_VALU", 0) / out["SQ_WAVES"]
    out["lds_insts_per_wave"] = out.get("SQ_INSTS_LDS", 0) / out["SQ_WAVES"]
if out.get("SQ_WAVE_CYCLES"):
    out["valu_active_over_wave_cycles"] = out.get("SQ_ACTIVE_INST_VALU", 0) / out["SQ_WAVE_CYCLES"]
    out["lds_active_over_wave_cycles"] = out.get("SQ_ACTIVE_INST_LDS", 0) / out["SQ_WAVE_CYCLES"]
    out["lds_bank_conflict_over_wave_cycles"] = out.get("SQ_LDS_BANK_CONFLICT", 0) / out["SQ_WAVE_CYCLES"]
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:      # KiB on gfx950
    out["hbm_bytes_per_env_step_counter"] = (out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024 / 8192
    out["algorithmic_bytes_per_env_step"] = 2 * 1088 + 1536 + 24 + 67 * 4
json.dump(out, open("gpurun_out/pmc_hands_%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out -path "*pmch_${TAG}_*" -name "*.csv" -size +6M -delete; find gpurun_out -path "*pmch_${TAG}_*" -name "*.db" -delete
