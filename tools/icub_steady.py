#!/usr/bin/env python
"""iCub push, auto-reset, random actions: step time and number of envs in the complex class (robot collision sphere within the contact margin of the object)
as the batch approaches its stationary mix.   python tools/icub_steady.py [--envs 32768] [--steps 1500] [--joint] [--max-steps 500]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32768); ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--joint", action="store_true"); ap.add_argument("--max-steps", type=int, default=500); ap.add_argument("--window", type=int, default=250)
ap.add_argument("--desync", action="store_true", help="de-synchronise the episode clocks after reset() (step counters U{0..max_steps-1}, as bench.py's headline "
                "does): after max_steps steps the batch holds episodes of every age and the windows stop alternating between episode halves")
ap.add_argument("--recycle", action="store_true", help="rounds 1-2 protocol: the same 8 action tensors over and over (a constant mean action per env: the "
                "commanded hand pose drifts into a workspace corner); default since round 3: fresh i.i.d. actions for every step")
args = ap.parse_args()
import numpy as np, torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import icub_table
import parity
tbl, model, info = icub_table("l")
ov = parity.icub_overrides(info, "l", 0 if args.joint else 1, 0, 1)
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=args.envs, robot=_capi.ROBOT_ICUB, flags=_capi.F_AUTO_RESET, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2,
                   max_steps=args.max_steps, **ov)
eng.reset()
if args.desync:
    st = eng.get_state()
    st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, args.max_steps, args.envs).astype(np.float32)
    eng.set_state(st)
dev = torch.device("cuda", 0)
torch.manual_seed(20260928)      # (the same action sequence in every run: A/B runs compare the same trajectories)
act = [torch.rand((args.envs, eng.act_dim), device=dev) * 2 - 1 for _ in range(8)]
out = torch.zeros((args.envs, eng.obs_dim + 2), device=dev)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
res = []
NP = min(args.window, 50)
for w in range(args.steps // args.window):
    if not args.recycle:                 # a window's actions: NP fresh tensors per chunk, generated outside the timed region below
        pass
    torch.cuda.synchronize(); t0 = time.perf_counter(); gen_s = 0.0
    for k in range(args.window):
        if not args.recycle and k % NP == 0:
            torch.cuda.synchronize(); g0 = time.perf_counter()
            act = [torch.rand((args.envs, eng.act_dim), device=dev) * 2 - 1 for _ in range(NP)]
            torch.cuda.synchronize(); gen_s += time.perf_counter() - g0
        eng.step_device(act[k % len(act)].data_ptr(), out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); el = time.perf_counter() - t0 - gen_s
    res.append({"steps_done": (w + 1) * args.window, "ms_per_step": round(el / args.window * 1e3, 4), "M_env_steps_per_s": round(args.envs * args.window / el / 1e6, 2),
                "complex_envs": eng.kernel_info()[5], "finite": bool(torch.isfinite(out).all())})
print(json.dumps({"actions": "recycled pool of 8" if args.recycle else "i.i.d. per step", "episode_clocks": "de-synchronised" if args.desync else "synchronised", "workload": "iCubPushGymEnv %s, %d envs, auto-reset, max_steps %d" % ("joint control" if args.joint else "IK position control", args.envs, args.max_steps),
                  "lane": os.environ.get("PBRE_ICUB_LANE", "1"), "windows": res}))
