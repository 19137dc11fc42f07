#!/usr/bin/env python
"""Long-horizon soak on one MI355X: continuous rollouts with in-kernel auto-reset and random actions; every state and output
must stay finite and physically plausible (objects on the table, unit quaternions, joints inside their limits + margin)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table, icub_table
import parity

dev = torch.device("cuda", 0)


def soak(name, eng, steps, chunk, act_dim, lim_lo, lim_hi):
    n, nd, xo = eng.num_envs, eng.ndof, eng.x_off
    eng.reset()
    out = torch.zeros((n, eng.obs_dim + 2), device=dev)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    episodes = 0
    t0 = time.perf_counter()
    worst = 0.0
    for c in range(steps // chunk):
        act = torch.rand((chunk, n, act_dim), device=dev, generator=gen) * 2 - 1
        for k in range(chunk):
            eng.step_device(act[k].data_ptr(), out.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all()), "non-finite output"
        st = eng.get_state()
        assert np.isfinite(st).all(), "non-finite state"
        q = st[:, :nd]
        if lim_lo is not None:      # joint control clips its targets to the limits; IK targets are not limit-aware (nor are the reference's)
            assert (q > lim_lo - 0.2).all() and (q < lim_hi + 0.2).all(), "joint far outside its limits"
        assert np.abs(np.linalg.norm(st[:, nd + 3:nd + 7], axis=1) - 1).max() < 1e-4, "object quaternion not normalised"
        z = st[:, nd + 2]
        worst = max(worst, float(np.abs(z - 0.65).max()))
        episodes = int(st[:, xo + 5].sum())
    el = time.perf_counter() - t0
    print(json.dumps({"soak": name, "envs": n, "steps": steps, "env_steps_per_s": n * steps / el, "episodes_completed": episodes,
                      "max_abs_object_height_dev_m": worst, "complex_envs_at_end": eng.kernel_info()[5]}))


tbl, _ = panda_table()
low = np.array([-2.9671, -1.8326, -2.9671, -3.1416, -2.9671, -0.0873, -2.9671, 0.0, 0.0]); high = np.array([2.9671, 1.8326, 2.9671, 0.0, 2.9671, 3.8223, 2.9671, 0.04, 0.04])
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=131072, flags=_capi.F_AUTO_RESET, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, max_steps=1000)
soak("panda push, auto-reset", eng, int(os.environ.get("SOAK_STEPS", "30000")), 500, 7, low, high)
del eng
tbl, model, info = icub_table("l")
ov = parity.icub_overrides(info, "l", 1, 0, 1)
lo = np.array([l["lower"] for l in model["links"] if l["jtype"]]); hi = np.array([l["upper"] for l in model["links"] if l["jtype"]])
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=32768, robot=_capi.ROBOT_ICUB, flags=_capi.F_AUTO_RESET, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2,
                   max_steps=500, **ov)
# IK mode: with random Cartesian actions the commanded hand pose wanders to unreachable corners of the workspace and the IK
# solution leaves the joint ranges; the position motors (impulse cap 1e5 N * dt) overpower the limit rows (cap 100), as they
# would in Bullet with the same defaults -- only finiteness is asserted here
soak("icub push (IK), auto-reset", eng, int(os.environ.get("SOAK_STEPS_ICUB", "3000")), 250, 3, None, None)
