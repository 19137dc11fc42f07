"""Golden vectors captured from the reference's own Gym classes (tools/make_golden.py: reference Python glue
executed over the oracle's physics) pin the glue restated in oracle/ and in the Python env layer: observation
order/limits, float32-limit scaling, reward, termination and counter logic, start poses.  CPU only."""
import os

import numpy as np
import pytest

import orc
from pybullet_robot_envs.envs import utils
from pybullet_robot_envs.envs import pandaPushGymEnv, pandaReachGymEnv, pandaPushGymGoalEnv

# bounds on the scaled observation of one replayed step (measured on the lane emulation: Panda 2.2e-4 / 7.6e-7, iCub 4.2e-6 / 4.6e-7)
TOL_VEL, TOL_OTHER = 1.5e-3, 5e-6
WORST = {"vel": 0.0, "other": 0.0}


def check_scaled_obs(ob, ref, tag, k, tol_vel=None, tol_other=None):
    """Scaled observation of one step from the reference's state (fp32 engine vs the fp64 capture), per group: the three
    end-effector velocity entries (6:9; the Panda's are divided by 0.03..0.07 before the Box scaling) and everything else."""
    e = np.abs(np.asarray(ob) - ref)
    ev, eo = e[6:9].max(), np.delete(e, [6, 7, 8]).max()
    WORST["vel"] = max(WORST["vel"], ev); WORST["other"] = max(WORST["other"], eo)
    assert ev < (tol_vel or TOL_VEL) and eo < (tol_other or TOL_OTHER), (tag, k, ev, eo)


G = np.load(os.path.join(os.path.dirname(__file__), "golden", "panda_glue.npz"))

CASES = [("pushA", 1, 1000), ("pushB", 1, 6), ("reachC", 0, 5), ("goalD", 2, 4), ("goalE", 2, 4), ("ikF", 1, 1000), ("repR", 1, 7)]


@pytest.mark.parametrize("tag,task,max_steps", CASES)
def test_oracle_glue_reproduces_reference(panda, tag, task, max_steps):
    o = orc.Oracle(panda["table"], task=task)
    o.task.max_steps = max_steps
    if tag.startswith("ik"):
        o.set_ik_mode()
    if tag.startswith("rep"):
        o.task.action_repeat = 3
    pre, act = G[tag + "_pre_state"], G[tag + "_actions"]
    for k in range(len(act)):
        st, out = o.batch_step(pre[k:k + 1], act[k:k + 1])
        raw = out[0, :-2]
        assert np.abs(raw - G[tag + "_raw_obs"][k]).max() < 1e-12          # float64 glue: identical up to rounding order
        assert abs(out[0, -2] - G[tag + "_reward"][k]) < 1e-9
        assert out[0, -1] == G[tag + "_done"][k]
        assert st[0, 35] == G[tag + "_counter"][k]
        if k + 1 < len(act):                                               # the physics under the reference classes was the oracle
            assert np.abs(st[0, :31] - G[tag + "_pre_state"][k + 1][:31]).max() < 1e-12
            if tag.startswith("ik"):                                       # commanded hand pose: accumulate, scale, clip
                assert np.abs(st[0, 38:44] - G[tag + "_pre_state"][k + 1][38:44]).max() < 1e-12


def test_reset_start_poses(panda):
    o = orc.Oracle(panda["table"], task=1)
    st, obs = o.batch_reset(1)
    assert np.abs(st[0, :35] - G["pushA_reset_state"][:35]).max() < 1e-12   # K3/K6: settled cube, default target
    assert np.allclose(G["obj_init_pose"], [0.45, 0.0, 0.695, 0, 0, 0.3826834323650898, 0.9238795325112867], atol=1e-15)
    assert G["h_table"] == 0.625
    o.set_ik_mode()
    st, obs = o.batch_reset(1)
    assert np.abs(st[0, :31] - G["ikF_reset_state"][:31]).max() < 1e-12     # IK-mode reset: one IK solve + one extra step
    assert np.abs(st[0, 38:44] - G["ikF_reset_state"][38:44]).max() < 1e-12


@pytest.mark.parametrize("cls,tag", [(pandaPushGymEnv, "push"), (pandaReachGymEnv, "reach"), (pandaPushGymGoalEnv, "goal")])
def test_spaces_bit_identical(emu_lib, cls, tag):
    env = cls(_lib=emu_lib)
    box = env.observation_space["observation"] if tag == "goal" else env.observation_space
    assert box.low.dtype == np.float32 and box.low.tobytes() == G[tag + "_obs_low"].tobytes()
    assert box.high.tobytes() == G[tag + "_obs_high"].tobytes()
    assert env.action_space.low.tobytes() == G[tag + "_act_low"].tobytes()
    assert env.action_space.high.tobytes() == G[tag + "_act_high"].tobytes()
    assert env.action_space.shape == (7,)
    if tag == "push":
        env = cls(_lib=emu_lib, use_IK=1)
        assert env.observation_space.low.tobytes() == G["ik_obs_low"].tobytes()
        assert env.observation_space.high.tobytes() == G["ik_obs_high"].tobytes()
        assert env.action_space.low.tobytes() == G["ik_act_low"].tobytes() and env.action_space.shape == (6,)


def test_utils_bit_identical(emu_lib):
    env = pandaPushGymEnv(_lib=emu_lib)
    box = env.observation_space
    for x, s, u in zip(G["utils_x"], G["utils_scaled"], G["utils_unscaled"]):
        assert utils.scale_gym_data(box, x).tobytes() == s.tobytes()
        assert utils.unscale_gym_data(box, s).tobytes() == u.tobytes()
    assert utils.goal_distance(G["utils_a"], G["utils_b"]).tobytes() == G["utils_dist"].tobytes()
    assert np.allclose(G["k5_scale"], [-0.14285712, 0.0, -0.49767446], atol=1e-8)
    assert abs(utils.goal_distance(np.array([.45, 0, .695]), np.array([.5, .05, .695])) - 0.07071067811865475) < 1e-16


@pytest.mark.parametrize("cls,tag,kw", [
    (pandaPushGymEnv, "pushA", {}), (pandaPushGymEnv, "pushB", {"max_steps": 6}),
    (pandaReachGymEnv, "reachC", {"max_steps": 5}), (pandaPushGymGoalEnv, "goalD", {"max_steps": 4, "tg_pose_rnd_std": 0.0}),
    (pandaPushGymEnv, "ikF", {"use_IK": 1}), (pandaPushGymEnv, "repR", {"action_repeat": 3, "max_steps": 7})])
def test_env_classes_match_reference_outputs(emu_lib, cls, tag, kw):
    """The drop-in classes (fp32 device algorithm, run through the CPU lane emulation here) return what the
    reference classes returned: scaled obs, reward, done -- step by step from the reference's own states."""
    env = cls(_lib=emu_lib, **kw)
    goal = tag.startswith("goal")
    o = env.reset()
    o = o["observation"] if goal else o
    if tag != "goalD":
        assert np.abs(o - G[tag + "_reset_obs"]).max() < 2e-3
    pre, act = G[tag + "_pre_state"], G[tag + "_actions"]
    for k in range(len(act)):
        s = np.zeros((1, 48), np.float32)
        s[0] = pre[k]
        env._engine.set_state(s)
        ob, r, d, info = env.step(act[k])
        ob = ob["observation"] if goal else ob
        assert ob.dtype == np.float64 and ob.shape == G[tag + "_obs"][k].shape
        check_scaled_obs(ob, G[tag + "_obs"][k], tag, k)
        assert abs(float(r) - G[tag + "_reward"][k]) < 2e-5 * max(1, abs(G[tag + "_reward"][k]))
        assert float(d) == G[tag + "_done"][k]
        assert int(env._env_step_counter) == G[tag + "_counter"][k]
        if goal:
            assert bool(info["is_success"]) == bool(G[tag + "_success"][k])
