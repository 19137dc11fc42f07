#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_icub.py -q -s -k "at_size" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-300
