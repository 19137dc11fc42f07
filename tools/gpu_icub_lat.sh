#!/bin/bash
export TMPDIR=/tmp
for L in 0 1; do for N in 128 2048; do for M in "" "--joint"; do
  PBRE_ICUB_LANE=$L timeout 300 python tools/bench_icub.py --envs $N --steps 20 $M 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LANE=$L', d['workload'][:55], 'ms/step %.3f kernel_ms %.3f' % (d['ms_per_step'], d['kernel_ms']), {k:d[k] for k in d if 'complex' in k or 'contact' in k})"
done; done; done
