#!/bin/bash
# iCub: step latency of the two engines at batch sizes that leave most SIMDs empty (a wave's chain, not throughput), hold protocol and stationary mix
export TMPDIR=/tmp
for L in 0 1; do for N in 2048 8192; do for M in "" "--joint"; do
  PBRE_ICUB_LANE=$L timeout 300 python tools/icub_steady.py --desync --envs $N --steps 1000 --window 250 $M 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LANE=$L', d['workload'][:60], [(w['steps_done'], w['ms_per_step'], w['complex_envs']) for w in d['windows']])"
done; done; done
