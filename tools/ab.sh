#!/bin/bash
# A/B of fast-kernel register limits: bench each libpbre_w*.so variant
for w in 1 2 3 4; do
  echo "== waves/SIMD limit $w"
  PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_w$w.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.1fM env-steps/s  ms/step %.4f kernel_ms %.4f'%(d['value']/1e6,d['ms_per_step'],d['roofline']['kernel_ms']))"
done
