#!/bin/bash
# lane-per-env iCub path: parity test details, then A/B runs
C=$(pwd)/pybullet-robot-envs_amd/csrc
timeout 1200 python -m pytest tests/test_gpu_icub.py -x -q 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-250
b() { timeout 300 python tools/bench_icub.py "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-50s %7.1f M  %.3f ms  kernel %.3f' % (d['workload'], d['env_steps_per_s']/1e6, d['ms_per_step'], d['kernel_ms']))"; }
echo "== default lib, joint, 32768: iters 150 / 76 / 2"
b --envs 32768 --joint; b --envs 32768 --joint --iters 76; b --envs 32768 --joint --iters 2
echo "== nofallback"
PBRE_LIB=$C/libpbre_nofb.so b --envs 32768 --joint
PBRE_LIB=$C/libpbre_nofb.so b --envs 65536 --joint
echo "== mreg0 (3 waves / CU)"
PBRE_LIB=$C/libpbre_mreg0.so b --envs 32768 --joint
PBRE_LIB=$C/libpbre_mreg0.so b --envs 49152 --joint
echo "== 16384 envs (1 wave per CU)"
b --envs 16384 --joint
PBRE_LIB=$C/libpbre_nofb.so b --envs 16384 --joint
