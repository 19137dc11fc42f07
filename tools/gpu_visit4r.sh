#!/bin/bash
export TMPDIR=/tmp
PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_nocas.so python - <<'PY' 2>&1 | grep "^{"
import json, os, sys, time
sys.path.insert(0, "pybullet-robot-envs_amd")
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
for n in (4096, 16384):
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, use_ik=1)
    eng.reset()
    t0 = time.perf_counter(); eng.reset(); t = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"lib": "nocas", "envs": n, "use_ik": 1, "reset_ms": t, "info": eng.kernel_info()}), flush=True)
PY
python - <<'PY' 2>&1 | grep "^{"
import json, os, sys, time
sys.path.insert(0, "pybullet-robot-envs_amd")
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
for n in (4096, 16384):
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, use_ik=1)
    eng.reset()
    t0 = time.perf_counter(); eng.reset(); t = (time.perf_counter() - t0) * 1e3
    info = eng.kernel_info()
    a = np.zeros((n, eng.act_dim), np.float32)
    t0 = time.perf_counter()
    for _ in range(20): eng.step(a)
    ts = (time.perf_counter() - t0) / 20 * 1e3
    print(json.dumps({"lib": "main", "envs": n, "use_ik": 1, "reset_ms": t, "info": info, "step_ms_host_path": ts, "info_after": eng.kernel_info()}), flush=True)
PY
