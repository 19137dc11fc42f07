#!/bin/bash
TAG=${1:-r02e}
mkdir -p gpurun_out; export TMPDIR=/tmp; ROOTDIR=$(pwd)
short() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f=d.get('fresh_reset') or {}
print('%s: steady %.1f M (%.4f ms)  fresh %.1f M (%.4f ms)  ratio %.3f  k_fast %.4f ms  complex/step %.1f' % (sys.argv[2], d['value']/1e6, d['ms_per_step'], f.get('value',0)/1e6, f.get('ms_per_step',0), d['value']/max(f.get('value',1),1), d['roofline']['kernel_ms'], d['config'].get('complex_envs_per_step_timed_region_rank0',-1)))
" "$1" "$2"; }
run() { # name, lib suffix, env assignments...
  NAME=$1; V=$2; shift; shift
  LIB=$ROOTDIR/pybullet-robot-envs_amd/csrc/libpbre${V:+_$V}.so
  env PBRE_LIB=$LIB "$@" timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-host-path 2>/dev/null | tail -1 > gpurun_out/${TAG}_ab_$NAME.json
  short gpurun_out/${TAG}_ab_$NAME.json "$NAME"
}
for rep in 1 2; do
  run prio3_$rep "" A=1
  run prio0_$rep prio0 A=1
  run prio3_fastfirst_$rep "" PBRE_RC_FIRST_MIN=1000000
  run prio0_fastfirst_$rep prio0 PBRE_RC_FIRST_MIN=1000000
done
echo "== kernel trace (steady state, prio3)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/trace_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-other-configs --no-host-path --no-fresh --steps 20 > /dev/null 2>&1)
f=$(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1); python tools/trace_steps.py $f 10
echo "== kernel trace (steady state, prio0)"
(cd /tmp && PBRE_LIB=$ROOTDIR/pybullet-robot-envs_amd/csrc/libpbre_prio0.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOTDIR/gpurun_out/trace0_$TAG -o run -- python $ROOTDIR/bench.py --no-cpu-baseline --no-other-configs --no-host-path --no-fresh --steps 20 > /dev/null 2>&1)
f=$(find gpurun_out/trace0_$TAG -name "*kernel_trace.csv" | head -1); python tools/trace_steps.py $f 10
find gpurun_out/trace_$TAG gpurun_out/trace0_$TAG -name "*.csv" -size +4M -delete; find gpurun_out -name "*.db" -delete
