#!/usr/bin/env python
"""Throughput of the 64-lane iCub engine on one MI355X (extra measurement, not the headline bench):
iCubPushGymEnv config (IK position control of the left hand, or joint control), N envs, device-resident actions.
    python tools/bench_icub.py [--envs 32768] [--steps 20] [--joint]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32768)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--joint", action="store_true")
ap.add_argument("--warm", type=int, default=1500, help="untimed steps with ZERO actions before the timed ones (the robot holds its post-reset pose, "
                "so the timed steps still start from the reset state): the GPU's clocks settle for THIS load.  0 = the old burst protocol "
                "(3 warm-up steps right after reset(): measures whatever clock state the reset left behind)")
ap.add_argument("--ik-iters", type=int, default=0, help="cap of the IK iterations (default: the reference's 100)")
ap.add_argument("--iters", type=int, default=0, help="solver iterations (default: the reference's 150); 2 isolates everything but the solver loop")
args = ap.parse_args()

import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import icub_table
import parity

tbl, model, info = icub_table("l")
ov = parity.icub_overrides(info, "l", 0 if args.joint else 1, 0, 1)
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=args.envs, robot=_capi.ROBOT_ICUB, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2,
                   **(dict(phys={"solver_iters": args.iters}) if args.iters else {}), **(dict(ik_max_iters=args.ik_iters) if args.ik_iters else {}), **ov)
t0 = time.perf_counter()
eng.reset()
t_reset = time.perf_counter() - t0
dev = torch.device("cuda", 0)
act = [torch.rand((args.envs, eng.act_dim), device=dev) * 2 - 1 for _ in range(4)]
out = torch.zeros((args.envs, eng.obs_dim + 2), device=dev)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
zero = torch.zeros((args.envs, eng.act_dim), device=dev)
for k in range(args.warm):
    eng.step_device(zero.data_ptr(), out.data_ptr(), s.cuda_stream)
for k in range(3):
    eng.step_device(act[k % 4].data_ptr(), out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(args.steps):
    eng.step_device(act[k % 4].data_ptr(), out.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"workload": "iCubPushGymEnv %s, %d envs" % ("joint control" if args.joint else "IK position control", args.envs),
                  "env_steps_per_s": args.envs * args.steps / el, "ms_per_step": el / args.steps * 1e3,
                  "kernel_ms": eng.timing()[3], "reset_s": t_reset, "warm_zero_action_steps": args.warm, "vgprs": eng.kernel_info()[1], "complex_envs": eng.kernel_info()[5], "finite": bool(torch.isfinite(out).all())}))
