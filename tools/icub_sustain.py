#!/usr/bin/env python
"""iCub push, joint control with ZERO actions (the robot holds its pose: every step costs the same, no contacts develop): ms per step over
consecutive windows of a long run -- does the post-reset rate hold under sustained load, or does it depend on the GPU's clock state?
    python tools/icub_sustain.py [--envs 32768] [--windows 10] [--window 2000]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--envs", type=int, default=32768); ap.add_argument("--windows", type=int, default=10); ap.add_argument("--window", type=int, default=2000)
args = ap.parse_args()
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import icub_table
import parity
tbl, model, info = icub_table("l")
ov = parity.icub_overrides(info, "l", 0, 0, 1)
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=args.envs, robot=_capi.ROBOT_ICUB, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, max_steps=10 ** 9, **ov)
eng.reset()
dev = torch.device("cuda", 0)
act = torch.zeros((args.envs, eng.act_dim), device=dev)
out = torch.zeros((args.envs, eng.obs_dim + 2), device=dev)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
res = []
for w in range(args.windows):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(args.window):
        eng.step_device(act.data_ptr(), out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); res.append(round((time.perf_counter() - t0) / args.window * 1e3, 4))
print(json.dumps({"workload": "iCub push, joint control, zero actions, %d envs" % args.envs, "lane": os.environ.get("PBRE_ICUB_LANE", "default"),
                  "ms_per_step_per_window_of_%d" % args.window: res, "complex_envs": eng.kernel_info()[5], "finite": bool(torch.isfinite(out).all())}))
