#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
for V in varb varc; do echo "== $V"; PBRE_LIB=$C/libpbre_$V.so timeout 300 python tools/rt_rc_probe.py 16 2>&1 | grep -E "differ|engine" | cut -c1-200; done
PROBE_TORCH=0 timeout 600 python tools/host_async_probe.py 2>&1 | grep -vE "amdgpu.ids" | grep -E "pipelined|synchronous Engine.step:" | cut -c1-400 | tee gpurun_out/r06n_host_async_probe.txt
