#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=pybullet-robot-envs_amd/csrc
for N in 16384 131072; do timeout 600 python tools/rt_ab.py $N $C/libpbre.so 2>&1 | grep -vE "amdgpu.ids"; done | tee gpurun_out/r06zb_rt_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "residual or threshold" 2>&1 | tail -5 | tee -a gpurun_out/r06zb_rt_ab.txt
