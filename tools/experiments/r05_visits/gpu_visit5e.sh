#!/bin/bash
# Round 5, visit E: which of the two chain changes costs time?  Same box, stationary step at 16384 / 131072 envs, four builds of pbre_capi.
export TMPDIR=/tmp
D=$(pwd)/pybullet-robot-envs_amd/csrc
for r in 1 2; do
for V in "" _r4chain _freeonly _nroonly; do
  echo "--- libpbre$V"; PBRE_LIB=$D/libpbre$V.so timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-40,130-260
done; done | tee gpurun_out/r05e_chain_ab4.txt
