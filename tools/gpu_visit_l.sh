#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_icub.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-250
b() { timeout 300 python tools/bench_icub.py "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s %7.1f M  %.3f ms  kernel %.3f complex %s' % (d['workload'], d['env_steps_per_s']/1e6, d['ms_per_step'], d['kernel_ms'], d.get('complex_envs')))"; }
b --envs 32768 --joint; b --envs 32768; b --envs 65536; b --envs 131072
bash tools/prof_icub.sh j32 --envs 32768 --steps 100 --joint 2>&1 | grep -E "kw_dyn|kw_quad|kw_fin|kw_obj"
