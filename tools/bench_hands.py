#!/usr/bin/env python
"""Throughput of the iCub-with-hands engine (BASELINE config 5 stand-in: 60 simulated DoF, multi-finger contacts) on one
MI355X -- extra measurement, not the headline bench.  Every env runs the reference's grasp demo
(examples/helloworlds/helloworld_icub.py:61-95) opening: pre_grasp, hand above the object; the hand is then lowered until the
palm presses on the object and the fingers close (grasp, force 10); the timed loop keeps commanding hand poses around that
pose (IK, device-resident actions), so every env carries robot-object and object-table contacts (--press, the round-2
workload).  Default since round 3: the whole scripted grasp up to the lift, the timed steps hold the brick in the air.  Reports env-steps/s, the step kernel's duration and how many envs have fingertip contacts in the timed region.
    python tools/bench_hands.py [--envs 8192] [--steps 20] [--joint]"""
import argparse
import json
import math as m
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--joint", action="store_true", help="joint control (37 absolute targets) instead of IK hand poses")
ap.add_argument("--press", action="store_true", help="the round-2 workload: palm pressing on the resting object")
args = ap.parse_args()

import numpy as np
import torch
from pybullet_robot_envs import _client
from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv
from demo_icub_hands import quat

cid = _client.connect(args.envs)
t0 = time.perf_counter()
robot = iCubHandsEnv(cid, use_IK=0 if args.joint else 1, control_arm='r')
t_reset = time.perf_counter() - t0
eng = robot._engine
pos_cl = [0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 1.57, 0.8, 0.5, 0.8]
dev = torch.device("cuda", 0)
if args.joint:
    robot.grasp(pos_cl)
    home = torch.tensor(np.asarray(robot.sim_home(), np.float32)[robot.controlled_dofs()], device=dev)
    act = [home + (torch.rand((args.envs, eng.act_dim), device=dev) - 0.5) * 0.4 for _ in range(4)]
elif args.press:
    # (the round-2 workload) hand above the object, palm down (demo phase 1), then lowered until the palm presses on the object,
    # fingers closing: sustained robot-object contacts + the four object-table contacts in every env
    q1 = quat([0, 0, m.pi / 2])
    robot.pre_grasp(); robot.step_simulation(10)
    robot.apply_action([0.5, -0.03, 0.72] + q1); robot.pre_grasp(); robot.step_simulation(60)
    robot.apply_action([0.5, -0.03, 0.69] + q1); robot.pre_grasp(); robot.step_simulation(40)
    robot.grasp(pos_cl); robot.step_simulation(20)
    base = torch.tensor([0.5, -0.03, 0.69, 0.0, 0.0, m.pi / 2], dtype=torch.float32, device=dev)
    jit = torch.tensor([0.004, 0.004, 0.002, 0.01, 0.01, 0.01], device=dev)
    act = [base + (torch.rand((args.envs, 6), device=dev) - 0.5) * jit for _ in range(4)]
else:
    # BASELINE config 5: the demo's scripted grasp (helloworld_icub.py:61-107) up to the lift; the timed steps hold the brick in
    # the air (same workload as bench.py's other_configs): 3-4 fingertips on the brick, no object-table contact
    e2 = [m.pi / 2, m.pi / 3, -m.pi]
    robot.pre_grasp(); robot.step_simulation(10)
    robot.apply_action([0.49, 0.0, 0.8] + quat([0, 0, m.pi / 2]), max_vel=5); robot.pre_grasp(); robot.step_simulation(60)
    robot.apply_action([0.485, 0.0, 0.72] + quat(e2), max_vel=5); robot.pre_grasp(); robot.step_simulation(60)
    robot.grasp(pos_cl); robot.step_simulation(60)
    robot.apply_action([0.45, 0, 0.9] + quat(e2), max_vel=5); robot.grasp(pos_cl); robot.step_simulation(60)
    lift = float(np.atleast_2d(np.asarray(robot.get_object_pose()))[:, 2].mean()) - 0.65
    hp = eng.get_state()[0][eng.x_off + 6:eng.x_off + 12].astype("float32")          # the commanded pose as the engine holds it (Euler)
    base = torch.tensor(hp, dtype=torch.float32, device=dev)
    jit = torch.tensor([0.004, 0.004, 0.002, 0.0, 0.0, 0.0], device=dev)
    act = [base + (torch.rand((args.envs, 6), device=dev) - 0.5) * jit for _ in range(4)]
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
print(json.dumps({"workload": "iCubHandsEnv %s, %d envs" % ("joint control (37 targets)" if args.joint else ("palm pressing on the object, fingers closing, IK hand-pose control" if args.press else "scripted grasp, brick held in the air, IK hand-pose control"), args.envs),
                  "env_steps_per_s": args.envs * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                  "kernel_ms": eng.timing()[3], "reset_s": t_reset, "vgprs": eng.kernel_info()[1],
                  "envs_with_fingertip_contact": int((tail[:, 5] > 0).sum()), "mean_contact_points": float(tail[:, 6].mean()),
                  "brick_lift_m": None if (args.joint or args.press) else lift,
                  "mean_fingertips_in_contact": float(tail[:, 5].mean()), "finite": bool(torch.isfinite(out).all())}))
