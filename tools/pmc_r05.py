#!/usr/bin/env python
"""rocprofv3 --pmc counter_collection CSVs of tools/profile_r05.sh (round 3: profile_r03.sh) -> gpurun_out/<tag>_pmc_hbm.json and <tag>_pmc_sq.json.
FETCH_SIZE / WRITE_SIZE unit on gfx950: KiB (the guide's HBM section).  Calibration as in round 2 (profiles/r02_pmc_hbm.json): the
lane-per-env kernels' 4-byte loads with a 192-byte lane stride read 1.0x the counter (k_classify reads exactly one 64-B line per env),
coalesced 64-B rows (k_observe) need the guide's x2 FETCH_SIZE correction; WRITE_SIZE as reported.
Per kernel the mean over the launches of the LAST THIRD of the run (the stationary steps: the counters of the first launches after
reset(), when no env is complex and the 2-waves-per-SIMD variant runs, are reported separately)."""
import collections, csv, glob, json, re, sys

tag, n_envs = sys.argv[1], int(sys.argv[2])


def load(name, counters):
    f = glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, name), recursive=True)
    if not f:
        return {}
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        c = r.get("Counter_Name")
        if c in counters:
            m = re.match(r"(?:void )?(?:pbre::)?(k\w+(?:<[\w, ]+>)?)", r["Kernel_Name"])
            if m:
                # (round 5: the step kernels carry a trailing bool, the solver-residual-threshold variant; `false` is the default build and
                # keeps the names of the earlier rounds, `true` is marked RT)
                name = m.group(1)
                if name.startswith("k_fused<"):      # k_fused<MODE, PAIR, RT, TAIL> (round 5: PAIR -- the simple envs' waves' mapping; round 6: RT, TAIL -- the instantiation with tail pairs)
                    a = [x.strip() for x in name[len("k_fused<"):-1].split(",")] + ["false"] * 3
                    name = "k_fused<" + a[0] + (", pair" if a[1] == "true" else "") + (", RT" if a[2] == "true" else "") + (", tail" if a[3] == "true" else "") + ">"
                name = name.replace(", false>", ">").replace(", true>", ", RT>")
                per[name][c].append(float(r["Counter_Value"]))
    return per


def summarise(per):
    out = {}
    for k, d in per.items():
        out[k] = {}
        for c, v in d.items():
            tail = v[len(v) * 2 // 3:] or v
            out[k][c] = {"calls": len(v), "mean_all": sum(v) / len(v), "mean_stationary_third": sum(tail) / len(tail)}
    return out


fs, ws = summarise(load("FETCH_SIZE", {"FETCH_SIZE"})), summarise(load("WRITE_SIZE", {"WRITE_SIZE"}))
hbm = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_r03.sh), bench.py stationary protocol, %d envs, 1 MI355X" % n_envs,
       "raw_kib": {"FETCH_SIZE": fs, "WRITE_SIZE": ws}}
for kf in ("k_fused<7>", "k_fused<7, tail>", "k_fused<7, pair>", "k_fast<7, 3>", "k_fast<7, 2>", "k_fast_pair<7>"):
    if kf in fs and kf in ws:
        f_, w_ = fs[kf]["FETCH_SIZE"]["mean_stationary_third"] * 1024, ws[kf]["WRITE_SIZE"]["mean_stationary_third"] * 1024
        hbm[kf] = {"envs_per_launch": n_envs, "fetch_bytes": f_, "write_bytes": w_, "hbm_bytes": f_ + w_, "hbm_bytes_per_env_step": (f_ + w_) / n_envs,
                   "algorithmic_bytes_per_env_step": 444, "launches": fs[kf]["FETCH_SIZE"]["calls"]}
# the variant that steps the stationary batch is the headline's dominant kernel
dom = "k_fast<7, 3>" if ("k_fast<7, 3>" in hbm and hbm["k_fast<7, 3>"]["launches"] >= hbm.get("k_fast<7, 2>", {"launches": 0})["launches"]) else "k_fast<7, 2>"
if "k_fast_pair<7>" in hbm and dom not in hbm:
    dom = "k_fast_pair<7>"          # a batch the pair kernel steps (<= 65536 envs)
for kf in ("k_fused<7, pair>", "k_fused<7>", "k_fused<7, tail>"):      # the step as one launch (round 5): that kernel IS the step
    if kf in hbm and hbm[kf]["launches"] >= hbm.get(dom, {"launches": 0})["launches"]:
        dom = kf
if dom in hbm:
    hbm["k_fast<7>"] = dict(hbm[dom], variant=dom)
    hbm["step_kernel"] = dict(hbm[dom], variant=dom)
# the two-kernel step (PBRE_FUSED=0 passes): k_fast's own traffic
fs0, ws0 = summarise(load("FETCH_SIZE0", {"FETCH_SIZE"})), summarise(load("WRITE_SIZE0", {"WRITE_SIZE"}))
two = {}
for kf in ("k_fast<7, 3>", "k_fast<7, 2>", "k_fast_pair<7>", "k_row_list<7>"):
    if kf in fs0 and kf in ws0:
        f_, w_ = fs0[kf]["FETCH_SIZE"]["mean_stationary_third"] * 1024, ws0[kf]["WRITE_SIZE"]["mean_stationary_third"] * 1024
        two[kf] = {"fetch_bytes": f_, "write_bytes": w_, "hbm_bytes_per_env_step_of_the_batch": (f_ + w_) / n_envs, "launches": fs0[kf]["FETCH_SIZE"]["calls"]}
if two:
    hbm["two_kernel_step_PBRE_FUSED_0"] = two
hbm["calibration"] = {"k_classify_fetch_bytes_expected": 64 * n_envs,
                      "k_classify_fetch_bytes_counter": fs.get("k_classify", {}).get("FETCH_SIZE", {"mean_all": 0})["mean_all"] * 1024,
                      "k_observe_fetch_bytes_expected": 192 * n_envs,
                      "k_observe_fetch_bytes_counter_x2": 2 * fs.get("k_observe", {}).get("FETCH_SIZE", {"mean_all": 0})["mean_all"] * 1024}
json.dump(hbm, open("gpurun_out/%s_pmc_hbm.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in hbm.items() if k.startswith("k_fast") or k.startswith("k_fused") or k.startswith("two_")}, indent=1))

sq = load("SQ", {"SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"})
out = {}
for k, d in sq.items():
    if not (k.startswith("k_fast<7") or k.startswith("k_fast_pair<7") or k.startswith("k_row_list<7") or k.startswith("k_fast_rc<7") or k.startswith("k_fused<7")):
        continue
    o = {c: (sum(v[len(v) * 2 // 3:]) / max(1, len(v[len(v) * 2 // 3:]))) for c, v in d.items()}
    o["launches"] = len(next(iter(d.values())))
    if o.get("SQ_WAVE_CYCLES"):
        o["valu_active_over_wave_cycles"] = o.get("SQ_ACTIVE_INST_VALU", 0) / o["SQ_WAVE_CYCLES"]
        o["wait_any_over_wave_cycles"] = o.get("SQ_WAIT_ANY", 0) / o["SQ_WAVE_CYCLES"]
        o["wait_inst_any_over_wave_cycles"] = o.get("SQ_WAIT_INST_ANY", 0) / o["SQ_WAVE_CYCLES"]
    if o.get("SQ_WAVES"):
        o["valu_insts_per_wave"] = o.get("SQ_INSTS_VALU", 0) / o["SQ_WAVES"]
    out[k] = o
if "k_fast<7, 3>" not in out and "k_fast<7, 2>" not in out and "k_fast_pair<7>" in out:
    out["k_fast<7>"] = dict(out["k_fast_pair<7>"], variant="k_fast_pair<7>")
if "k_fast<7, 3>" in out or "k_fast<7, 2>" in out:
    d3, d2 = out.get("k_fast<7, 3>", {"launches": 0}), out.get("k_fast<7, 2>", {"launches": 0})
    out["k_fast<7>"] = dict(d3 if d3["launches"] >= d2["launches"] else d2, variant="k_fast<7, 3>" if d3["launches"] >= d2["launches"] else "k_fast<7, 2>")
for kf in ("k_fused<7, pair>", "k_fused<7>", "k_fused<7, tail>"):
    if kf in out and out[kf]["launches"] >= out.get("k_fast<7>", {"launches": 0})["launches"]:
        out["k_fast<7>"] = dict(out[kf], variant=kf)
        out["step_kernel"] = dict(out[kf], variant=kf, note="one launch per step: the complex envs' row waves (a few hundred, ~100 k instructions each, plus the idle "
                                  "row blocks that exit at once) are in SQ_WAVES and SQ_INSTS_VALU beside the simple envs' waves; PBRE_FUSED=0 passes give k_fast's own")
sq0 = load("SQ0", {"SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"})
two = {}
for k, d in sq0.items():
    if not (k.startswith("k_fast<7") or k.startswith("k_fast_pair<7") or k.startswith("k_row_list<7")):
        continue
    o = {c: (sum(v[len(v) * 2 // 3:]) / max(1, len(v[len(v) * 2 // 3:]))) for c, v in d.items()}
    o["launches"] = len(next(iter(d.values())))
    if o.get("SQ_WAVE_CYCLES"):
        o["valu_active_over_wave_cycles"] = o.get("SQ_ACTIVE_INST_VALU", 0) / o["SQ_WAVE_CYCLES"]
    if o.get("SQ_WAVES"):
        o["valu_insts_per_wave"] = o.get("SQ_INSTS_VALU", 0) / o["SQ_WAVES"]
    two[k] = o
if two:
    out["two_kernel_step_PBRE_FUSED_0"] = two
json.dump(out, open("gpurun_out/%s_pmc_sq.json" % tag, "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("valu_insts_per_wave", "valu_active_over_wave_cycles", "wait_any_over_wave_cycles", "launches") if kk in v} for k, v in out.items() if "SQ_WAVES" in v or "launches" in v}, indent=1))
print(json.dumps({k: {kk: v.get(kk) for kk in ("valu_insts_per_wave", "launches")} for k, v in out.get("two_kernel_step_PBRE_FUSED_0", {}).items()}))
