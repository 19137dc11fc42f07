#!/usr/bin/env python
"""Copy a rocprofv3 kernel_stats.csv into profiles/ with kernel names truncated to 100 chars."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.reader(open(src)))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:100]
        w.writerow(r)
