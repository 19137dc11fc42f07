#!/bin/bash
# A/B of the k_fast_rc scheduling threshold: fresh-start and mid-episode throughput
for t in 1 256 2048 1000000000; do
  PBRE_RC_FIRST_MIN=$t python bench.py --steps 50 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('thr', $t, 'fresh %.1f M (%.3f ms)' % (d['value']/1e6, d['ms_per_step']), 'steady %.1f M' % (d['steady_state']['value']/1e6), 'kernel %.3f pair %.3f' % (d['roofline']['kernel_ms'], d['roofline']['step_launch_pair_ms']))"
done
