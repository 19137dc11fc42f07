#!/usr/bin/env python
"""k_fast time split: kernel time at 131072 envs for several PGS iteration counts (the intercept is the non-loop part)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np, torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
n = 131072
dev = torch.device("cuda", 0)
for iters in (2, 50, 150):
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, phys=dict(solver_iters=iters))
    eng.reset()
    act = torch.rand((16, n, 7), device=dev) * 2 - 1
    out = torch.zeros((n, eng.obs_dim + 2), device=dev)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    for k in range(16):
        eng.step_device(act[k].data_ptr(), out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    print("iters %3d: k_fast %.4f ms" % (iters, eng.timing()[3]))
