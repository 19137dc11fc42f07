#!/bin/bash
# VALU instructions per wave of k_fast for whole source trees of earlier commits unpacked + built under gpurun_tmp/<commit>/
ROOTDIR=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for S in "$@"; do
  T=$ROOTDIR/gpurun_tmp/$S
  rm -rf gpurun_out/vb_$S
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $ROOTDIR/gpurun_out/vb_$S -o run -- python $T/bench.py --envs 131072 --no-cpu-baseline --no-other-configs --no-host-path --no-fresh --steps 10 --warmup 2 --preroll 300 > $ROOTDIR/gpurun_out/vb_$S.log 2>&1)
  python - $S <<'PY'
import csv, glob, re, sys, collections
s = sys.argv[1]
f = glob.glob("gpurun_out/vb_%s/**/*counter_collection.csv" % s, recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    m = re.match(r"(?:void )?(?:pbre::)?(k_fast\w*<[\d, ]+>|k_row_list<7>)", r["Kernel_Name"])
    if m:
        per[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(per.items()):
    w, v, c = (sum(d[x]) for x in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"))
    print("%-8s %-18s launches %4d  VALU/wave %8.0f  cycles/wave %8.0f" % (s, k, len(d["SQ_WAVES"]), v / max(w, 1), c / max(w, 1)))
PY
  tail -n 2 gpurun_out/vb_$S.log | cut -c1-200
  find gpurun_out/vb_$S -name "*.csv" -size +2M -delete; find gpurun_out/vb_$S -name "*.db" -delete
done
