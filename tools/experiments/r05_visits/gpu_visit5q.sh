#!/bin/bash
# Round 5, visit Q: explicit sweeps before the object block's closed form: 12 (libpbre_oc12.so) against 22 (default), one box; a lane whose
# bound fails drags its wave through the remaining explicit sweeps, so any loss of acceptance shows as a slower step
export TMPDIR=/tmp
D=$(pwd)/pybullet-robot-envs_amd/csrc
for r in 1 2; do
for V in _oc12 ""; do
  echo "--- libpbre$V"; PBRE_LIB=$D/libpbre$V.so timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
done; done | tee gpurun_out/r05q_oc_k_ab.txt
PBRE_LIB=$D/libpbre_oc12.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "closed_form or reset_and_steps or rollout or one_launch" 2>&1 | grep -vE "^/opt/amdgpu" | tail -3
