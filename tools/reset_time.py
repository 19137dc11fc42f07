#!/usr/bin/env python
"""Wall time of a full pbre_reset (reset_simulation: 100 + 101 settle steps of every env) by batch size."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
for n in (4096, 16384, 131072):
    for use_ik in (0, 1):
        eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, use_ik=use_ik)
        eng.reset()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); eng.reset(); ts.append((time.perf_counter() - t0) * 1e3)
        st = eng.get_state()
        print(json.dumps({"envs": n, "use_ik": use_ik, "reset_ms": [round(t, 2) for t in ts], "state_checksum": float(np.abs(st[:, :31]).sum())}), flush=True)
        eng.close()
