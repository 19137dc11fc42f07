#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for N in 131072 16384; do
bash tools/trace_panda_steady3.sh $N r04m_$N 2>&1 | tail -14
done
