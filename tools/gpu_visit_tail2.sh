#!/bin/bash
# GPU visit: the suite on the current build, then same-box A/B: previous build / tail pairs off / by the hint / forced counts
TAG=${1:-r06r}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-400
for r in 1 2; do
  for v in "old:1" "cur:0" "cur:1" "cur:120" "cur:180" "cur:300"; do
    lib=${v%%:*}; tp=${v#*:}
    if [ $lib = old ]; then export PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_old.so; else unset PBRE_LIB; fi
    echo "--- lib=$lib PBRE_TAIL_PAIR=$tp"
    PBRE_TAIL_PAIR=$tp timeout 300 python tools/tail_probe.py --sizes 16384,131072 --preroll 1100 --steps 600 2>&1 | grep envs | cut -c1-260
  done
done 2>&1 | tee gpurun_out/${TAG}_ab.txt
