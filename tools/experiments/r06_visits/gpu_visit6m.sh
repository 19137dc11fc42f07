#!/bin/bash
export TMPDIR=/tmp
for F in 16 0 8; do for U in 1 0; do PBRE_FUSED=$U timeout 300 python tools/rt_rc_probe.py $F 2>&1 | grep -vE "amdgpu.ids" | cut -c1-400; done; done
