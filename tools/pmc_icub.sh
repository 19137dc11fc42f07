#!/bin/bash
# SQ issue counters of the iCub's solver kernel (kw_quad: four lanes per env, pbre_lane.hip; joint control) in two --pmc passes.
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
run() { local name=$1; shift
    (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $ROOTDIR/gpurun_out/pmci_${TAG}_$name -o run -- python $ROOTDIR/tools/bench_icub.py --envs 32768 --steps 10 --joint > $ROOTDIR/gpurun_out/pmci_${TAG}_$name.log 2>&1)
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run b SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
python - $TAG <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {"source": "rocprofv3 --pmc (2 passes, tools/pmc_icub.sh), tools/bench_icub.py --envs 32768 --steps 10 --joint, 1 MI355X", "kernel": "kw_quad (16 envs per wave)"}
for name in ("a", "b"):
    fs = glob.glob("gpurun_out/pmci_%s_%s/**/*counter_collection.csv" % (tag, name), recursive=True)
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.rstrip().endswith("kw_quad"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items(): out[c] = sum(v) / len(v)
w = out.get("SQ_WAVES", 0); wc = out.get("SQ_WAVE_CYCLES", 0)
if w:
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"):
        if c in out: out[c.lower() + "_per_wave"] = out[c] / w
if wc:
    for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
        if c in out: out[c.lower() + "_over_wave_cycles"] = out[c] / wc
json.dump(out, open("gpurun_out/pmc_icub_%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out -path "*pmci_${TAG}_*" -name "*.csv" -size +6M -delete; find gpurun_out -path "*pmci_${TAG}_*" -name "*.db" -delete
