#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_icub.py -q -s -k "push_policy" 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
