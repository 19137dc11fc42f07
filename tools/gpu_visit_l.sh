#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_icub.py tests/test_gpu_hands.py -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/bench_icub.py --envs 4096 --steps 50 2>&1 | tail -1 | cut -c1-160
timeout 300 python tools/bench_icub.py --envs 8192 --steps 50 2>&1 | tail -1 | cut -c1-160
