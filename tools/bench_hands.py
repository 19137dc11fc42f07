#!/usr/bin/env python
"""Throughput of the iCub-with-hands engine (BASELINE config 5 stand-in: 60 simulated DoF, grasp scenario with fingertip
contacts) on one MI355X -- extra measurement, not the headline bench.  Scenario per env: reset, pre_grasp, the hand moves above
the object (joint targets), grasp(force 10); the timed loop then keeps commanding absolute joint targets around that pose with
the fingers closing on the object.  Device-resident actions.
    python tools/bench_hands.py [--envs 8192] [--steps 20] [--ik]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--ik", action="store_true")
args = ap.parse_args()

import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import icub_hands_table, GRASP_POS
import parity

tbl, model, info = icub_hands_table("r")
ov = parity.hands_overrides(info, "r", 1 if args.ik else 0)
eng = _capi.Engine(tbl, task=_capi.TASK_REACH, num_envs=args.envs, robot=_capi.ROBOT_ICUB_HANDS, obj_pose_rnd_std=0.02, **ov)
t0 = time.perf_counter()
eng.reset()
t_reset = time.perf_counter() - t0
eng.set_motors(info["fingers"], GRASP_POS, 0.1, 10.0)
dev = torch.device("cuda", 0)
if args.ik:
    base = torch.tensor([0.3, -0.1, 0.8, 0.0, 0.0, 1.0], device=dev)
    act = [base + (torch.rand((args.envs, 6), device=dev) - 0.5) * 0.04 for _ in range(4)]
else:
    home = torch.tensor(np.asarray(info["home"], np.float32)[info["controlled"]], device=dev)
    act = [home + (torch.rand((args.envs, eng.act_dim), device=dev) - 0.5) * 0.4 for _ in range(4)]
out = torch.zeros((args.envs, eng.obs_dim + 2), device=dev)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
for k in range(3):
    eng.step_device(act[k % 4].data_ptr(), out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(args.steps):
    eng.step_device(act[k % 4].data_ptr(), out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
el = time.perf_counter() - t0
tail = out[:, -9:-2]
print(json.dumps({"workload": "iCubHandsEnv %s, %d envs" % ("IK pose control" if args.ik else "joint control (37 targets)", args.envs),
                  "env_steps_per_s": args.envs * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                  "kernel_ms": eng.timing()[3], "reset_s": t_reset, "vgprs": eng.kernel_info()[1],
                  "envs_with_fingertip_contact": int((tail[:, 5] > 0).sum()), "finite": bool(torch.isfinite(out).all())}))
