#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=pybullet-robot-envs_amd/csrc
for N in 16384 131072; do timeout 900 python tools/rt_ab.py $N $C/libpbre.so $C/libpbre_rtfree0.so $C/libpbre_free2.so $C/libpbre.so 2>&1 | grep -vE "amdgpu.ids"; done | tee gpurun_out/r06zc_rt_ab.txt
