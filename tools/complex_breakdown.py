#!/usr/bin/env python
"""What makes the steady-state "complex" envs complex?  After the bench's pre-roll (auto-reset batch, i.i.d. actions) count the envs
whose state has a joint at / beyond a limit (limit rows only) against all complex envs (limit rows and / or robot contacts)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, model = panda_table()
n = 131072
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
eng.reset()
dev = torch.device("cuda", 0)
out = torch.zeros((n, eng.obs_dim + 2), device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
act = torch.empty((n, 7), device=dev)
lo = np.array([l["lower"] for l in model["links"] if l["jtype"]], np.float32)
hi = np.array([l["upper"] for l in model["links"] if l["jtype"]], np.float32)
res = []
for k in range(1, 1501):
    act.uniform_(-1, 1, generator=gen)
    eng.step_device(act.data_ptr(), out.data_ptr(), _capi.torch_stream(dev))
    if k in (250, 500, 750, 1000, 1050, 1250, 1500):
        torch.cuda.synchronize()
        c = eng.kernel_info()[5]
        q = eng.get_state_cols(0, 9)
        at = ((q <= lo) | (q >= hi))
        res.append({"step": k, "complex": int(c), "envs_with_joint_at_limit": int(at.any(1).sum()), "per_joint": at.sum(0).tolist(),
                    "done_last_step": int(out[:, -1].sum().item())})
        print(res[-1])
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "complex_breakdown.json"), "w"), indent=1)
