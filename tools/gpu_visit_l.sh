#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_icub.py -q 2>&1 | tail -2
for M in "" "--joint"; do timeout 600 python tools/icub_steady.py --envs 32768 --steps 1000 --window 250 $M 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'],[ (w['ms_per_step'],w['complex_envs']) for w in d['windows']])"; done
timeout 900 python tools/icub_push_soak.py --envs 32768 --steps 900 2>&1 | tail -1 | cut -c1-200
