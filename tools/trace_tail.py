#!/usr/bin/env python
"""per-kernel average duration over the last N launches of each kernel in a rocprofv3 kernel trace CSV, and the wall time per step
(first launch of the marker kernel to the next).   python tools/trace_tail.py <run_kernel_trace.csv> [--last 100] [--marker kw_fin]"""
import argparse, csv, collections
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--last", type=int, default=100); ap.add_argument("--marker", default="kw_fin")
a = ap.parse_args()
rows = sorted(csv.DictReader(open(a.csv)), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("pbre::", "")
    by[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
mk = [k for k in by if a.marker in k]
if mk:
    st = [s for s, e in by[mk[0]]][-a.last - 1:]
    print("step period over the last %d steps: %.1f us" % (len(st) - 1, (st[-1] - st[0]) / 1e3 / max(len(st) - 1, 1)))
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1][-a.last:])):
    t = v[-a.last:]
    if len(v) < 10: continue
    print("%-28s launches %6d   avg of last %d: %8.1f us" % (k[:28], len(v), len(t), sum(e - s for s, e in t) / len(t) / 1e3))
