"""Residual exit, simple class: WHY does a lane not take the closed form (and with it drag its whole wave onto the explicit rows)?  CPU emulation,
bench.py's stationary protocol in small; counts the lanes by reason mask (1 object rows not through after OC_K sweeps, 2 a motor may clamp, 4 trial
residuals not decreasing, 8 object bound) and records the slowest-moving cube among the lanes of each mask -- is "the cube moves" a usable predictor?
   usage: python tools/rt_why_probe.py [envs=1024] [preroll=500] [steps=150]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 500
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
eng = _capi.Engine(tbl, task=1, num_envs=n, lib=lib, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
eng.reset()
st = eng.get_state(); st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32); eng.set_state(st)
rng = np.random.default_rng(3)
for k in range(pre):
    eng.step(rng.uniform(-1, 1, (n, 7)).astype(np.float32))
eng.set_physics(solver_residual_threshold=1e-7)
why = (C.c_long * 16)(); sp = (C.c_float * 32)()
lib.pbre_emu_rt_why(why, sp, 1)
moving = 0
for k in range(steps):
    s = eng.get_state()
    v = np.linalg.norm(s[:, 25:28], axis=1); w = np.linalg.norm(s[:, 28:31], axis=1)
    moving += int(((v > 1e-3) | (w > 1e-2)).sum())
    eng.step(rng.uniform(-1, 1, (n, 7)).astype(np.float32))
lib.pbre_emu_rt_why(why, sp, 0)
tot = sum(why)
print("simple-class lane-steps %d of %d env-steps; cubes with |v| > 1e-3 m/s or |w| > 1e-2 rad/s before the step: %d (%.4f)" % (tot, n * steps, moving, moving / (n * steps)))
for m in range(16):
    if why[m]:
        print("  mask %2d: %8d lanes (%.5f)   slowest cube among them: |v| %.3g m/s |w| %.3g rad/s" % (m, why[m], why[m] / tot, sp[2 * m], sp[2 * m + 1]))
