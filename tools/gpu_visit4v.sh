#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do for V in "" ro2; do
  L=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre${V:+_$V}.so
  PBRE_LIB=$L timeout 600 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep "^{" | sed "s/^{/{\"lib\": \"${V:-main (4 slots)}\", /" | tee -a gpurun_out/r04v_ro_slots_ab.json | cut -c1-300
done; done
