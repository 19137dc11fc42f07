#!/bin/bash
# Round-4 counter passes of the headline command (stationary protocol: 1000-step pre-roll, then the timed steps), one --pmc pass each,
# no tracing domains besides the kernel dispatch records rocprofv3 needs:  FETCH_SIZE, WRITE_SIZE (HBM traffic), SQ issue counters.
#   usage: tools/profile_r04.sh <tag> [envs]      -> gpurun_out/<tag>_pmc_hbm.json, gpurun_out/<tag>_pmc_sq.json
TAG=${1:-r04}; N=${2:-131072}
ROOTDIR=$(pwd); export TMPDIR=/tmp
run() {  # <name> <counters...>
  local NAME=$1; shift
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $ROOTDIR/gpurun_out/pmc_${TAG}_$NAME -o run -- python $ROOTDIR/bench.py --envs $N --no-cpu-baseline --no-other-configs --no-host-path --no-fresh --no-shards --steps 20 --warmup 3 > $ROOTDIR/gpurun_out/pmc_${TAG}_$NAME.log 2>&1)
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run SQ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
python tools/pmc_r04.py $TAG $N
find gpurun_out/pmc_${TAG}_* -name "*.csv" -size +6M -delete; find gpurun_out/pmc_${TAG}_* -name "*.db" -delete
