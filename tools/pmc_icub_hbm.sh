#!/bin/bash
# HBM traffic of the iCub pipeline's kernels (FETCH_SIZE / WRITE_SIZE, one --pmc pass each; counter unit on gfx950: KiB), joint control,
# 32768 envs, post-reset steps -> gpurun_out/pmc_icub_hbm_$1.json.   usage: tools/pmc_icub_hbm.sh <tag>
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $ROOTDIR/gpurun_out/pmcih_${TAG}_$C -o run -- python $ROOTDIR/tools/bench_icub.py --envs 32768 --steps 20 --joint > $ROOTDIR/gpurun_out/pmcih_${TAG}_$C.log 2>&1)
done
python - $TAG <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]; n = 32768
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_icub_hbm.sh), tools/bench_icub.py --envs 32768 --steps 20 --joint, 1 MI355X",
       "note": "bytes = counter x 1024; lane-per-env kernels read the counter as is (calibrated on the Panda engine, profiles/r02_pmc_hbm.json)", "kernels": {}}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmcih_%s_%s/**/*counter_collection.csv" % (tag, cname), recursive=True)
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != cname: continue
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("pbre::", "")
        if k in ("kw_dyn", "kw_quad", "kw_quad_rc", "kw_fin", "kw_obj"): agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[-20:]
        out["kernels"].setdefault(k, {})[cname.lower() + "_bytes_per_env_step"] = 1024 * sum(v) / len(v) / n
tot = sum(sum(d.values()) for d in out["kernels"].values())
out["hbm_bytes_per_env_step_all_kernels"] = tot
out["algorithmic_bytes_per_env_step"] = 584
json.dump(out, open("gpurun_out/pmc_icub_hbm_%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out -path "*pmcih_${TAG}_*" -name "*.csv" -size +6M -delete; find gpurun_out -path "*pmcih_${TAG}_*" -name "*.db" -delete
