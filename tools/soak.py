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
                      "max_abs_object_height_dev_m": worst, "complex_envs_at_end": eng.kernel_info()[5],
                      "nan_inf_guard_bad_env_steps": eng.kernel_info()[12]}))


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
del eng
if os.environ.get("SOAK_STEPS_HANDS", "1500") != "0":
    # iCub with hands: random absolute joint targets inside the joint ranges (37 controlled joints incl. the right hand's fingers),
    # every 250 steps a finger command (open / pre-grasp / grasp with force 10) to a random half of the envs; no episodes.
    # Arms flailing into the table are stopped by hard contacts (impulse bound 1e10) against position motors (bound 1e5 N * dt):
    # the joint-limit rows (bound 100) give way, as on the iCub in IK mode -- the excursion is reported, only finiteness asserted
    from pybullet_robot_envs.model.table import icub_hands_table, GRASP_POS
    tbl, model, info = icub_hands_table("r")
    ov = parity.hands_overrides(info, "r", 0)
    eng = _capi.Engine(tbl, task=_capi.TASK_REACH, num_envs=8192, robot=_capi.ROBOT_ICUB_HANDS, obj_pose_rnd_std=0.03, **ov)
    lo = np.array([l["lower"] for l in model["links"] if l["jtype"]]); hi = np.array([l["upper"] for l in model["links"] if l["jtype"]])
    n, nd = eng.num_envs, eng.ndof
    eng.reset()
    out = torch.zeros((n, eng.obs_dim + 2), device=dev)
    s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    c_lo = torch.tensor(lo[info["controlled"]], dtype=torch.float32, device=dev); c_hi = torch.tensor(hi[info["controlled"]], dtype=torch.float32, device=dev)
    home = torch.tensor(np.asarray(info["home"], np.float32)[info["controlled"]], device=dev)
    steps = int(os.environ.get("SOAK_STEPS_HANDS", "1500"))
    rng = np.random.default_rng(5)
    t0 = time.perf_counter(); contacts = 0; fmax = 0.0; worst_exc = 0.0; worst_joint = ""
    tgt = home.repeat(n, 1)
    for c in range(steps // 250):
        cmd = [np.zeros(20), np.array([0.0] * 16 + [1.57, 0, 0, 0]), np.array(GRASP_POS)][c % 3]
        eng.set_motors(info["fingers"], cmd, 0.1, 10.0 if c % 3 == 2 else 0.0, mask=(rng.random(n) < 0.5).astype(np.uint8))
        for k in range(250):
            if k % 25 == 0:     # a new random target (a slow random walk of the arm, clipped to the limits by the engine)
                tgt = torch.minimum(torch.maximum(tgt + (torch.rand(tgt.shape, device=dev, generator=gen) - 0.5) * 0.6, c_lo - 0.1), c_hi + 0.1)
            eng.step_device(tgt.data_ptr(), out.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all()), "non-finite output"
        st = eng.get_state()
        assert np.isfinite(st).all(), "non-finite state"
        q = st[:, :nd]
        exc = np.maximum(lo - q, q - hi).max(axis=0)        # worst excursion beyond the limits, per joint
        worst_exc = max(worst_exc, float(exc.max())); worst_joint = info["dof_names"][int(exc.argmax())] if exc.max() >= worst_exc else worst_joint
        assert np.abs(np.linalg.norm(st[:, nd + 3:nd + 7], axis=1) - 1).max() < 1e-4, "object quaternion not normalised"
        contacts += int((st[:, nd + 13] > 0).sum()); fmax = max(fmax, float(st[:, nd + 7:nd + 12].max()))
    el = time.perf_counter() - t0
    print(json.dumps({"soak": "icub with hands, random joint targets + finger commands", "envs": n, "steps": steps, "env_steps_per_s": n * steps / el,
                      "env_chunks_with_robot_object_contact": contacts, "max_fingertip_force_N": fmax,
                      "worst_joint_limit_excursion_rad": worst_exc, "worst_joint": worst_joint}))
