#!/bin/bash
# RT (residual exit) step: where the time goes.  Writes gpurun_out/r06y_rt_time.txt
export TMPDIR=/tmp
ROOTDIR=$(pwd)
mkdir -p gpurun_out
O=$ROOTDIR/gpurun_out/r06y_rt_time.txt
: > $O
for N in 131072 16384; do
  for F in 1 0; do
    rm -rf gpurun_out/prof_r06y
    (cd /tmp && PBRE_FUSED=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/prof_r06y -o run -- python $ROOTDIR/tools/rt_time_probe.py $N 200 2>&1 | grep -vE "amdgpu.ids|^W2|rocprofv3" >> $O)
    s=$(find gpurun_out/prof_r06y -name "*kernel_stats.csv" | head -1)
    [ -n "$s" ] && python tools/compact_stats.py $s gpurun_out/r06y_stats_${N}_$F.csv && head -9 gpurun_out/r06y_stats_${N}_$F.csv | cut -c1-200 >> $O
    echo >> $O
  done
done
rm -rf gpurun_out/prof_r06y
cat $O
