#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for V in "" nocas; do
  L=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre${V:+_$V}.so
  PBRE_LIB=$L timeout 600 python tools/tail_probe.py --sizes 16384,65536,131072 --preroll 1100 2>&1 | grep "^{" | sed "s/^{/{\"lib\": \"${V:-main}\", /" | tee -a gpurun_out/r04e_cas_ab.json | cut -c1-330
done; done
