#!/bin/bash
export TMPDIR=/tmp
ROOTDIR=$(pwd)
cat > /tmp/ikreset.py <<'PY'
import json, os, sys, time
sys.path.insert(0, os.path.join(os.environ["ROOTDIR"], "pybullet-robot-envs_amd"))
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
n = 16384
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, use_ik=1)
t0 = time.perf_counter(); eng.reset(); print("reset ms", (time.perf_counter() - t0) * 1e3)
PY
export ROOTDIR
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ik -o run -- python /tmp/ikreset.py 2>&1 | grep "reset ms")
f=$(find /tmp/prof_ik -name "*kernel_stats.csv" | head -1); python3 -c "import csv,sys; [print(r[0][:40], r[1], r[3], r[5], r[6]) for r in list(csv.reader(open(sys.argv[1])))[:7]]" $f
