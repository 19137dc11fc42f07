#!/bin/bash
echo "== profile joint 32768"; bash tools/prof_icub.sh j32 --envs 32768 --steps 50 --joint
echo "== profile IK 65536"; bash tools/prof_icub.sh ik64 --envs 65536 --steps 50
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_icub.py -q 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-250
