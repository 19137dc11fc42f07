#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for F in 0 1 2; do
PBRE_FAST3=$F timeout 600 python tools/tail_probe.py --sizes 131072 --preroll 1100 --steps 400 2>&1 | grep "^{" | sed "s/^{/{\"PBRE_FAST3\": $F, /" | tee -a gpurun_out/r04p_fast3.json | cut -c1-330
done
