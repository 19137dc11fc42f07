#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (tools/pmc.sh) -> profiles/<tag>_pmc_hbm.json.
Counter unit on gfx950: KiB (the guide's HBM section).  Calibration kernels: k_classify reads exactly one 64-B line per env
with the lane-per-env access pattern; k_observe reads whole coalesced rows (needs the guide's x2 FETCH_SIZE correction)."""
import collections
import csv
import glob
import json
import re
import sys

tag, n_envs = sys.argv[1], int(sys.argv[2])
raw = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, cname), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == cname:
            m = re.match(r"(?:void )?(?:pbre::)?(k\w+(?:<\d+>)?)", r["Kernel_Name"])
            if m:
                agg[m.group(1)].append(float(r["Counter_Value"]))
    raw[cname] = {k: {"calls": len(v), "mean_kb": sum(v) / len(v)} for k, v in agg.items()}
kf = "k_fast<7>"
fetch = raw["FETCH_SIZE"][kf]["mean_kb"] * 1024
write = raw["WRITE_SIZE"][kf]["mean_kb"] * 1024
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc.sh), bench.py --steps 20 --warmup 3, %d envs, 1 MI355X" % n_envs,
    "raw": raw,
    "calibration": {
        "k_classify_fetch_bytes_expected": 64 * n_envs,
        "k_classify_fetch_bytes_counter": raw["FETCH_SIZE"].get("k_classify", {"mean_kb": 0})["mean_kb"] * 1024,
        "k_observe_fetch_bytes_expected": 192 * n_envs,
        "k_observe_fetch_bytes_counter_x2": 2 * raw["FETCH_SIZE"].get("k_observe", {"mean_kb": 0})["mean_kb"] * 1024,
        "note": "lane-per-env kernels (4-byte loads, 192-byte lane stride) read 1.0x the counter (k_classify reads exactly one 64-B line "
                "per env); the 16-lane row kernels (64-B coalesced rows) need the guide's x2 FETCH_SIZE correction (k_observe). "
                "WRITE_SIZE is used as reported.",
    },
    kf: {"envs_per_launch": n_envs, "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write,
         "hbm_bytes_per_env_step": (fetch + write) / n_envs, "algorithmic_bytes_per_env_step": 444,
         "note": "552 B/env of padded record + output accesses, 60 B of solver start values parked in the record, and the register spills around the solver loop"},
}
json.dump(out, open("gpurun_out/%s_pmc_hbm.json" % tag, "w"), indent=1)
print(json.dumps(out[kf], indent=1))
