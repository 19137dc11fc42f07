#!/usr/bin/env python
"""Per-step timeline from a rocprofv3 --kernel-trace CSV of a bench run: for the last `n` launches of k_fast<7>, when the step's
kernels start and end relative to the first start (us).  usage: tools/trace_steps.py <kernel_trace.csv> [n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ev = []
for r in rows:
    k = r["Kernel_Name"]
    if ", true>" in k and "k_fused" not in k:        # (round 5: the solver-residual-threshold variants of the step kernels are not part of this timeline)
        continue
    name = "k_fast" if ("k_fast<7>" in k or "k_fast<7," in k or "k_fast_pair<7>" in k) else ("k_row_list" if "k_row_list<7" in k else ("k_fast_rc" if "k_fast_rc<7" in k else None))
    if "k_fused<7" in k:
        name = "k_fused"
    if name:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
# round 5: the step as one launch (k_fused: row-list blocks + k_fast / pair waves in one grid) -- durations and the idle time between launches
fu = [e for e in ev if e[2] == "k_fused"][-n:]
if len(fu) > n // 2:
    d = sorted((e[1] - e[0]) / 1e3 for e in fu)
    g = sorted((b[0] - a[1]) / 1e3 for a, b in zip(fu, fu[1:]) if b[0] - a[1] < 200000)
    for nm, v in (("k_fused", d), ("gap_to_previous_step", g)):
        print(nm, "min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f  mean %.1f  (n = %d)" % (v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], v[-1], sum(v) / len(v), len(v)))
    sys.exit(0)
# round 5: the stationary step launches k_fast as two half grids back to back (pbre_capi.hip launch_step, PBRE_FAST3=3): twice as many k_fast
# launches as row-list launches in the tail -> merge them in pairs (the pairing with the smaller gaps) into one [start of A, end of B] event
tail = ev[-6 * n:]
nf, nr = sum(e[2] == "k_fast" for e in tail), sum(e[2] != "k_fast" for e in tail)
if nr and nf > 1.6 * nr:
    fi = [i for i, e in enumerate(ev) if e[2] == "k_fast"][-(4 * n + 1):]
    def gaps(off):
        return sum(ev[fi[k + 1]][0] - ev[fi[k]][1] for k in range(off, len(fi) - 1, 2))
    off = 0 if gaps(0) <= gaps(1) else 1
    drop, halves = set(), []
    for k in range(off, len(fi) - 1, 2):
        a, b = ev[fi[k]], ev[fi[k + 1]]
        halves.append(((a[1] - a[0]) / 1e3, (b[1] - b[0]) / 1e3, (b[0] - a[1]) / 1e3))
        ev[fi[k]] = (a[0], b[1], "k_fast"); drop.add(fi[k + 1])
    drop.update(fi[:off]); drop.update(i for i in range(fi[0]) if ev[i][2] == "k_fast")
    ev = [e for i, e in enumerate(ev) if i not in drop]
    h = sorted(halves)
    print("k_fast in two half-grid launches: median first %.1f us, second %.1f us, gap %.1f us" % (sorted(x[0] for x in halves)[len(h) // 2], sorted(x[1] for x in halves)[len(h) // 2], sorted(x[2] for x in halves)[len(h) // 2]))
fast = [i for i, e in enumerate(ev) if e[2] == "k_fast"][-n:]
dur = collections.defaultdict(list)
prev_end = None
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
        # idle time between the steps: from the previous step's last kernel end to this step's first kernel start; and the lag of the later
        # of the two kernels' starts behind the earlier
        if prev_end is not None and t0 - prev_end < 200000: dur["gap_to_previous_step"].append((t0 - prev_end) / 1e3)
        dur["second_kernel_start_lag"].append(abs(c[0] - s0) / 1e3)
        prev_end = max(e0, c[1])
    print(line)
print({k: round(sum(v) / len(v), 1) for k, v in dur.items()})
for k, v in dur.items():      # distribution over the listed steps (us): min, quartiles, max
    v = sorted(v)
    print(k, "min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f  (n = %d)" % (v[0], v[len(v) // 4], v[len(v) // 2], v[3 * len(v) // 4], v[-1], len(v)))
