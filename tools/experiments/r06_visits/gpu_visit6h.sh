#!/bin/bash
# Round 6, visit H: the stamped hand-over + contention tests (3 runs) and the iCub suite; timeline of the pipelined host path (kernel + copy trace)
export TMPDIR=/tmp
mkdir -p gpurun_out
ROOTDIR=$(pwd)
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_contention.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | grep -E "rows differ|passed|failed" | cut -c1-300; done
timeout 900 python -m pytest tests/test_gpu_icub.py -m gpu -q 2>&1 | grep -vE "^/opt/amdgpu" | tail -3 | cut -c1-300
for V in 1 0; do PBRE_ASYNC_D2H=$V timeout 600 python tools/host_async_probe.py 2>&1 | grep -E "pipelined|synchronous"; done
rm -rf gpurun_out/prof_async
(cd /tmp && PROBE_WARM=20 PROBE_STEPS=12 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $ROOTDIR/gpurun_out/prof_async -o run -- python $ROOTDIR/tools/host_async_probe.py > $ROOTDIR/gpurun_out/prof_async.log 2>&1)
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("gpurun_out/prof_async/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40], r.get("Queue_Id", "")))
for f in glob.glob("gpurun_out/prof_async/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Kind", "")) , ""))
ev.sort()
# the last ~40 events before the end of the pipelined phase: find the last k_rows_out or D2H copy
idx = [i for i, e in enumerate(ev) if "k_rows_out" in e[2] or "DEVICE_TO_HOST" in e[2].upper()]
if idx:
    lo = max(0, idx[len(idx) // 2] - 20)
    t0 = ev[lo][0]
    for e in ev[lo:lo + 45]:
        print("%9.1f us  +%8.1f us  %s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
else:
    print("no copy events found", len(ev))
PY
find gpurun_out/prof_async -name "*.db" -delete; find gpurun_out/prof_async -name "*.csv" -size +4M -delete
