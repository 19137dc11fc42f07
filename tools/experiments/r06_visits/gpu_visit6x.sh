#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/small_batch_probe.py 2>&1 | grep -vE "amdgpu.ids" | tee gpurun_out/r06x_small_batch_probe.txt
