#!/bin/bash
# HBM traffic counters, one --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# Counter passes use no tracing domains other than the kernel dispatch records rocprofv3 needs for --pmc.
TAG=$1; shift
ROOTDIR=$(pwd); export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $ROOTDIR/gpurun_out/pmc_${TAG}_$C -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-other-configs --no-host-path --preroll 200 "$@" > $ROOTDIR/gpurun_out/pmc_${TAG}_$C.log 2>&1)
  f=$(find gpurun_out/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1)
  echo "== $C: $f"
  [ -n "$f" ] && python - "$f" $C <<'PY'
import csv, sys, collections
f, cname = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == cname:
        agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print("%-42s calls %4d  mean %12.1f  (x1024 = %.3e B)" % (k, len(v), sum(v) / len(v), 1024 * sum(v) / len(v)))
PY
  find gpurun_out/pmc_${TAG}_$C -name "*.csv" -size +6M -delete; find gpurun_out/pmc_${TAG}_$C -name "*.db" -delete
done
