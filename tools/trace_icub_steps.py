#!/usr/bin/env python
"""Per-step timeline of the iCub lane pipeline from a rocprofv3 --kernel-trace CSV: steps are delimited by kw_fin launches; for the last `n`
steps: every kernel's start / end relative to the step's first kernel start, the step's span, and the time in which NO kernel of ours runs
(idle: launch gaps, fork / join packets).   usage: tools/trace_icub_steps.py <kernel_trace.csv> [n] [steps to print]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
show = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ev = []
for r in rows:
    m = re.match(r"(?:void )?(?:pbre::)?(kw_\w+)", r["Kernel_Name"])
    if m:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
ev.sort()
fins = [i for i, e in enumerate(ev) if e[2] == "kw_fin"]
steps = []
for a, b in zip(fins[-n - 1:-1], fins[-n:]):
    steps.append(ev[a + 1:b + 1])
agg = collections.defaultdict(list)
for k, st in enumerate(steps):
    t0 = min(e[0] for e in st); t1 = max(e[1] for e in st)
    # union coverage
    cov = 0; cur_s, cur_e = None, None
    for s, e, _ in sorted(st):
        if cur_e is None or s > cur_e:
            if cur_e is not None: cov += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cov += cur_e - cur_s
    agg["span"].append((t1 - t0) / 1e3); agg["idle_inside_step"].append((t1 - t0 - cov) / 1e3)
    if k > 0:
        prev_end = max(e[1] for e in steps[k - 1]); agg["gap_to_previous_step"].append((t0 - prev_end) / 1e3)
    for s, e, nm in st:
        agg["dur " + nm].append((e - s) / 1e3); agg["start " + nm].append((s - t0) / 1e3); agg["end " + nm].append((e - t0) / 1e3)
    if k >= len(steps) - show:
        print("step", k, " ".join("%s %.0f..%.0f" % (nm.replace("kw_", ""), (s - t0) / 1e3, (e - t0) / 1e3) for s, e, nm in st))
for k in sorted(agg):
    v = sorted(agg[k])
    print("%-28s median %8.1f  mean %8.1f  p90 %8.1f  (n = %d)" % (k, v[len(v) // 2], sum(v) / len(v), v[len(v) * 9 // 10], len(v)))
