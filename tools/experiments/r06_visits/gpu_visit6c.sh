#!/bin/bash
# Round 6, visit C: (1) which change broke the iCub pipeline's contact envs: libpbre_noto.so = quad_step without the timeout select;
# (2) k_fast_pair register-limited to 4 waves per SIMD (libpbre_wps4.so) against the default, two-kernel step, pair forced at every size
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$(pwd)/pybullet-robot-envs_amd/csrc
echo "== default lib"; timeout 600 python -m pytest tests/test_gpu_icub.py -m gpu -q -x -k "crafted_contact_states and not round" 2>&1 | grep -vE "^/opt/amdgpu" | tail -2 | cut -c1-300
echo "== noto lib"; PBRE_LIB=$C/libpbre_noto.so timeout 600 python -m pytest tests/test_gpu_icub.py -m gpu -q -k "crafted_contact_states or identical_pipelines or closed_loop" 2>&1 | grep -vE "^/opt/amdgpu" | tail -4 | cut -c1-300
for r in 1 2; do
echo "--- default (fused)"; timeout 300 python tools/tail_probe.py --sizes 131072,65536 --preroll 1100 --steps 300 2>&1 | grep envs | cut -c1-260
echo "--- PBRE_FUSED=0 PBRE_PAIR=1 default lib"; PBRE_FUSED=0 PBRE_PAIR=1 timeout 300 python tools/tail_probe.py --sizes 131072,65536 --preroll 1100 --steps 300 2>&1 | grep envs | cut -c1-260
echo "--- PBRE_FUSED=0 PBRE_PAIR=1 wps4"; PBRE_LIB=$C/libpbre_wps4.so PBRE_FUSED=0 PBRE_PAIR=1 timeout 300 python tools/tail_probe.py --sizes 131072,65536 --preroll 1100 --steps 300 2>&1 | grep envs | cut -c1-260
echo "--- PBRE_FUSED=0 PBRE_PAIR=0 (k_fast alone)"; PBRE_FUSED=0 PBRE_PAIR=0 timeout 300 python tools/tail_probe.py --sizes 131072,65536 --preroll 1100 --steps 300 2>&1 | grep envs | cut -c1-260
done 2>&1 | tee gpurun_out/r06c_pair_wps4_ab.txt
