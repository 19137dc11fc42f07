#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl.py -q -x -s 2>&1 | grep -vE "^/opt/amdgpu" | tail -12 | cut -c1-400
