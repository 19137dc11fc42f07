#!/bin/bash
# Short GPU visit for the iCub pipeline: its tests, post-reset throughput, stationary mix, one traced step.   usage: bash tools/gpu_visit_l.sh
timeout 900 python -m pytest tests/test_gpu_icub.py -q 2>&1 | tail -2
b() { timeout 300 python tools/bench_icub.py "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s %8.3f M  %.3f ms kernel %.3f' % (d['workload'], d['env_steps_per_s']/1e6, d['ms_per_step'], d['kernel_ms']))"; }
b --envs 32768; b --envs 32768 --joint; b --envs 131072
PBRE_ICUB_TRACE=10 timeout 300 python tools/bench_icub.py --envs 32768 --steps 20 --joint 2>&1 | grep "iCub step" | head -2
for M in "" "--joint"; do timeout 600 python tools/icub_steady.py --envs 32768 --steps 750 --window 250 $M 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'],[ (w['ms_per_step'],w['complex_envs']) for w in d['windows']])"; done
