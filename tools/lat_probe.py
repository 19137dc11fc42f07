#!/usr/bin/env python
"""Single-wave latency probes on one MI355X: kernel time of a batch small enough that every wave has a SIMD to itself."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
rng = np.random.default_rng(0)
for name, n, flags in [("k_step rows (4 envs/wave)", 4096, _capi.F_FORCE_GENERAL), ("k_step rows", 256, _capi.F_FORCE_GENERAL),
                       ("k_fast lane-per-env", 65536, 0), ("k_fast lane-per-env", 4096, 0)]:
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, flags=flags, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    eng.reset()
    for k in range(12):
        eng.step(rng.uniform(-1, 1, (n, 7)).astype(np.float32))
    print(name, n, "envs: kernel %.3f ms" % eng.timing()[3], "VGPRs", eng.kernel_info()[:2])
