#!/usr/bin/env python
"""Host-buffer entry point (Engine.step, page-locked staging) at the BASELINE batch: ms per step and the split the library reports.
A/B knobs: PBRE_ZERO_COPY=1 (kernels access the page-locked host buffers directly), PBRE_HOST_NONCOHERENT=1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
n = 131072
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
eng.reset()
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (n, 7)).astype(np.float32) for _ in range(4)]
for k in range(5):
    eng.step(acts[k % 4], copy=False)
K = 40
t0 = time.perf_counter()
for k in range(K):
    o, r, d = eng.step(acts[k % 4], copy=False)
el = time.perf_counter() - t0
ms = eng.timing()
print("zero_copy=%s noncoherent=%s: %.3f ms/step -> %.1f M env-steps/s (h2d %.3f, kernels %.3f, d2h %.3f ms), rows finite %s, mean reward %.4f" % (
    os.environ.get("PBRE_ZERO_COPY", "0"), os.environ.get("PBRE_HOST_NONCOHERENT", "0"), el / K * 1e3, n * K / el / 1e6, ms[0], ms[1], ms[2], bool(np.isfinite(o).all()), float(r.mean())))
