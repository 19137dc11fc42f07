#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=pybullet-robot-envs_amd/csrc
for N in 131072 16384; do echo "== $N envs"; timeout 900 python tools/ab_identity.py $C/libpbre.so $C/libpbre_ocnofb.so $N 1200 2>&1 | grep -vE "amdgpu.ids"; done | tee gpurun_out/r06za_oc_fallback_ab.txt
