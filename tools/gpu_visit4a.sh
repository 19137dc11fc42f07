#!/bin/bash
# Round 4, visit a: the pair kernel (k_fast_pair) -- parity under both mappings, then small-shard step times with and without it.
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | sed -n 2,3p
echo "== pytest tests/test_gpu_parity.py (both mappings)"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/r04a_pytest_parity.log 2>&1; grep -vE "^/opt/amdgpu" gpurun_out/r04a_pytest_parity.log | tail -6 | cut -c1-300
echo "== small shards: PBRE_PAIR=0 / 1, fresh + stationary"
for P in 0 1; do
  PBRE_PAIR=$P timeout 600 python tools/tail_probe.py --sizes 4096,16384,32768,65536,131072 --preroll 1100 2>&1 | grep "^{" | sed "s/^{/{\"PBRE_PAIR\": $P, /" | tee -a gpurun_out/r04a_pair_shards.json | cut -c1-330
done
echo "== the same with 2 solver iterations in the timed region (non-loop part)"
for P in 0 1; do
  PBRE_PAIR=$P timeout 600 python tools/tail_probe.py --sizes 16384,65536 --preroll 1100 --iters 2 2>&1 | grep "^{" | sed "s/^{/{\"PBRE_PAIR\": $P, /" | tee -a gpurun_out/r04a_pair_shards_iters2.json | cut -c1-330
done
