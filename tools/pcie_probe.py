#!/usr/bin/env python
"""PCIe-inclusive rate of the host-buffer entry point pbre_step (actions up, [obs|reward|done] rows down, pageable numpy
buffers) at the BASELINE batch: never the bench `value`, recorded in DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
n = 131072
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
eng.reset()
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (n, 7)).astype(np.float32) for _ in range(4)]
for k in range(3):
    eng.step(acts[k % 4])
t0 = time.perf_counter()
K = 30
for k in range(K):
    eng.step(acts[k % 4])
el = time.perf_counter() - t0
ms = eng.timing()
print("pbre_step host path: %.3f ms/step -> %.1f M env-steps/s (h2d %.3f ms, kernels %.3f ms, d2h %.3f ms)" % (el / K * 1e3, n * K / el / 1e6, ms[0], ms[1], ms[2]))
