#!/usr/bin/env python
"""Per-step timeline from a rocprofv3 --kernel-trace CSV of a bench run: for the last `n` launches of k_fast<7>, when the step's
kernels start and end relative to the first start (us).  usage: tools/trace_steps.py <kernel_trace.csv> [n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ev = []
for r in rows:
    k = r["Kernel_Name"]
    if ", true>" in k:        # (round 5: the solver-residual-threshold variants of the step kernels are not part of this timeline)
        continue
    name = "k_fast" if ("k_fast<7>" in k or "k_fast<7," in k or "k_fast_pair<7>" in k) else ("k_row_list" if "k_row_list<7" in k else ("k_fast_rc" if "k_fast_rc<7" in k else None))
    if name:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
fast = [i for i, e in enumerate(ev) if e[2] == "k_fast"][-n:]
dur = collections.defaultdict(list)
for i in fast:
    s0, e0, _ = ev[i]
    # the complex-env kernel of the same step: the nearest k_row_list / k_fast_rc launch overlapping or just before
    cand = [e for e in ev[max(0, i - 3):i + 3] if e[2] != "k_fast" and e[1] > s0 - 50000 and e[0] < e0 + 50000]
    c = min(cand, key=lambda e: abs(e[0] - s0)) if cand else None
    t0 = min(s0, c[0]) if c else s0
    line = "k_fast %7.1f..%7.1f us" % ((s0 - t0) / 1e3, (e0 - t0) / 1e3)
    dur["k_fast"].append((e0 - s0) / 1e3)
    if c:
        line += "   %s %7.1f..%7.1f us" % (c[2], (c[0] - t0) / 1e3, (c[1] - t0) / 1e3)
        dur[c[2]].append((c[1] - c[0]) / 1e3)
        dur["span"].append((max(e0, c[1]) - t0) / 1e3)
    print(line)
print({k: round(sum(v) / len(v), 1) for k, v in dur.items()})
for k, v in dur.items():      # distribution over the listed steps (us): min, quartiles, max
    v = sorted(v)
    print(k, "min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f  (n = %d)" % (v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], v[-1], len(v)))
