#!/bin/bash
# host-buffer path of pbre_step: staged copies vs kernels accessing the page-locked buffers (PBRE_ZERO_COPY bit 0 actions, bit 1 rows)
for Z in 0 1 2 3; do
  PBRE_ZERO_COPY=$Z timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids | tail -1
done
