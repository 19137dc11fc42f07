#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
for N in 131072 16384; do timeout 600 python tools/ab_identity.py $C/libpbre.so $C/libpbre_nopk.so $N 1200 2>&1 | grep -vE "amdgpu.ids" | cut -c1-300; done | tee gpurun_out/r06w_pk_square_ab.txt
