#!/usr/bin/env python
"""iCub: the lane-per-env pipelines (PBRE_ICUB_LANE=1 quad pipeline, =2 one-kernel LDS variant) against the lane-group kernel (=0) on the
same seeded batch: per-step differences of the output rows and of the final states.  `--lib emu` runs the CPU lane emulation instead.
    python tools/lane_ab.py [--envs 256] [--steps 60] [--variants 1,2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=256)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--variants", default="1")
ap.add_argument("--lib", default="hip")
args = ap.parse_args()

import numpy as np
import parity
from pybullet_robot_envs import _capi

lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so")) if args.lib == "emu" else _capi.load()


def run(lane, task, arm, use_ik, ori, auto):
    os.environ["PBRE_ICUB_LANE"] = str(lane)
    eng, ora, info = parity.make_icub_pair(_capi.Engine, lib, args.envs, task=task, control_arm=arm, use_ik=use_ik, control_orientation=ori,
                                           obj_std=0.05, tg_std=0.1, max_steps=15, flags=(2 if auto else 0))
    eng.reset()
    rng = np.random.default_rng(3)
    outs = []
    for k in range(args.steps):
        a = rng.uniform(-1, 1, (args.envs, eng.act_dim)).astype(np.float32)
        ob, rw, dn = eng.step(a)
        outs.append(np.concatenate([ob, rw[:, None], dn[:, None]], 1))
    return np.array(outs), eng.get_state(), eng.kernel_info()


worst = 0.0
for cfg in [(0, "l", 1, 0), (1, "r", 1, 1), (1, "l", 0, 0), (2, "r", 1, 1)]:
    for auto in (False, True):
        b, sb, ib = run(0, *cfg, auto)
        for v in [int(x) for x in args.variants.split(",")]:
            a, sa, ia = run(v, *cfg, auto)
            d = np.abs(a - b) / (1 + np.abs(b))
            ds = np.abs(sa - sb) / (1 + np.abs(sb))
            first = int(np.argmax(d.reshape(d.shape[0], -1).max(1) > 1e-3)) if (d > 1e-3).any() else -1
            print("task %d arm %s ik %d ori %d auto %d variant %d: max rel diff rows %.2e (first step over 1e-3: %d) state %.2e  finite %s  info %s" % (
                *cfg, auto, v, d.max(), first, ds.max(), bool(np.isfinite(a).all()), ia[:6]))
            worst = max(worst, float(d.max()))
print("worst", worst)
