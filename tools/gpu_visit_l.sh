#!/bin/bash
run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>/tmp/err.log | tail -1 | python -c "
import sys,json; b=json.loads(sys.stdin.read()); r=b['other_configs']['icub_reach']; print('panda', round(b['ms_per_step'],4), 'icub', r['value']/1e6, r['ms_per_step'], r['ms_per_step_all_repetitions'], r['steady_random_actions']['ms_per_step'])"; }
for i in 1 2 3; do run --no-host-path; done
