"""iCub golden vectors captured from the reference's own classes (tools/make_golden.py: iCubReachGymEnv,
iCubPushGymEnv, iCubPushGymGoalEnv executed over the oracle's physics) pin the iCub glue restated in oracle/:
observation order (COM-frame hand pose, raw velocity, the 10 controlled joints), the IK-mode hand-pose accumulation
(scales 0.005 / 0.01 / 0.02, Euler and workspace clipping per arm), joint-mode targets, the three reward variants,
termination and counter logic.  CPU only."""
import os

import numpy as np
import pytest

import orc

# bounds on the scaled observation of one replayed step (measured on the lane emulation: Panda 2.2e-4 / 7.6e-7, iCub 4.2e-6 / 4.6e-7)
TOL_VEL, TOL_OTHER = 2e-4, 2e-5
WORST = {"vel": 0.0, "other": 0.0}


def check_scaled_obs(ob, ref, tag, k, tol_vel=None, tol_other=None):
    """Scaled observation of one step from the reference's state (fp32 engine vs the fp64 capture), per group: the three
    end-effector velocity entries (6:9; the Panda's are divided by 0.03..0.07 before the Box scaling) and everything else."""
    e = np.abs(np.asarray(ob) - ref)
    ev, eo = e[6:9].max(), np.delete(e, [6, 7, 8]).max()
    WORST["vel"] = max(WORST["vel"], ev); WORST["other"] = max(WORST["other"], eo)
    assert ev < (tol_vel or TOL_VEL) and eo < (tol_other or TOL_OTHER), (tag, k, ev, eo)


G = np.load(os.path.join(os.path.dirname(__file__), "golden", "icub_glue.npz"))

#        tag      arm task use_ik ori max_steps reward_type
CASES = [("reachG", "l", 0, 1, 0, 5, 1), ("pushH", "l", 1, 1, 0, 1000, 0), ("pushI", "r", 1, 1, 1, 6, 1),
         ("pushJ", "l", 1, 0, 0, 1000, 1), ("goalK", "r", 2, 1, 1, 4, 1), ("goalL", "r", 2, 1, 1, 4, 1), ("repS", "l", 0, 1, 0, 4, 1)]


def make(arm, task, use_ik, ori, max_steps, reward_type):
    o, tbl, info = orc.icub_oracle(arm, task=task, use_ik=use_ik, control_orientation=ori)
    o.task.max_steps = max_steps
    o.task.reward_type = reward_type
    # the object of the scene: the reach env's default is duck_vhacd, the push envs' cube_small (icub_reach_gym_env.py:32,
    # icub_push_gym_env.py:32), both simulated as the box stand-ins of model/objects.py
    from pybullet_robot_envs.model.objects import object_physics
    orc.set_object(o, object_physics("duck_vhacd" if task == 0 else "cube_small"))
    return o, info


@pytest.mark.parametrize("tag,arm,task,use_ik,ori,max_steps,reward_type", CASES)
def test_oracle_glue_reproduces_reference(tag, arm, task, use_ik, ori, max_steps, reward_type):
    o, info = make(arm, task, use_ik, ori, max_steps, reward_type)
    if tag.startswith("rep"):
        o.task.action_repeat = 2
    pre, act = G[tag + "_pre_state"], G[tag + "_actions"]
    ox = o.state_floats - 16
    assert pre.shape[1] == o.state_floats == 80 and act.shape[1] == o.task.n_act
    for k in range(len(act)):
        st, out = o.batch_step(pre[k:k + 1], act[k:k + 1])
        assert np.abs(out[0, :-2] - G[tag + "_raw_obs"][k]).max() < 1e-12
        assert abs(out[0, -2] - G[tag + "_reward"][k]) < 1e-9
        assert out[0, -1] == G[tag + "_done"][k]
        assert st[0, ox + 3] == G[tag + "_counter"][k]
        if k + 1 < len(act):
            nxt = G[tag + "_pre_state"][k + 1]
            assert np.abs(st[0, :ox] - nxt[:ox]).max() < 1e-12
            if use_ik:
                assert np.abs(st[0, ox + 6:ox + 12] - nxt[ox + 6:ox + 12]).max() < 1e-12


def test_pruned_legs_are_exact():
    """The engine's iCub model drops the legs (limbs rooted at the fixed base are independent dynamical systems and no env
    observes them): every output of the 20-DoF model equals the full 32-DoF model's, bit for bit (fp64 oracle)."""
    from pybullet_robot_envs.model.table import icub_table

    def mk(full):
        tbl, model, info = icub_table("l", full=full)
        o = orc.Oracle(tbl, task=1)
        o.set_icub(info, 1, "l", 1, 0)
        return o
    a, b = mk(False), mk(True)
    assert (a.ndof, b.ndof) == (20, 32)
    sa, oa = a.batch_reset(1)
    sb, ob = b.batch_reset(1)
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(0)
    for k in range(12):
        act = rng.uniform(-1, 1, (1, 3))
        sa, ra = a.batch_step(sa, act)
        sb, rb = b.batch_step(sb, act)
        assert np.array_equal(ra, rb)


def test_reset_and_bookkeeping():
    o, info = make("l", 0, 1, 0, 5, 1)
    st, obs = o.batch_reset(1)
    ref = G["reachG_reset_state"]
    ox = o.state_floats - 16
    assert np.abs(st[0, :ox] - ref[:ox]).max() < 1e-12                 # IK at reset + 1 + 100 + 101 steps, settled object
    assert np.abs(st[0, ox + 6:ox + 12] - ref[ox + 6:ox + 12]).max() < 1e-12
    # joint bookkeeping the reference derives by name (icub_env.py:107-150)
    # the reference numbers joints over the full SDF model (38 links, 32 DoF); the engine simulates it without the legs
    from pybullet_robot_envs.model.table import icub_table
    _, full, fl = icub_table("l", full=True)
    _, _, fr = icub_table("r", full=True)
    jidx = [i for i, l in enumerate(full["links"]) if l["jtype"] != 0]
    assert [jidx[d] for d in fl["controlled"]] == list(G["joints_to_control_l"])
    assert [jidx[d] for d in fr["controlled"]] == list(G["joints_to_control_r"])
    assert [fl["ee_link"], fr["ee_link"]] == list(G["end_eff_idx"])
    assert [fl["dof_names"][d] for d in fl["controlled"]] == [info["dof_names"][d] for d in info["controlled"]]
    assert np.allclose(G["home_hand_pose_l"], o.task.home_hand_pose[:]) 
    o2, _ = make("r", 1, 1, 1, 6, 1)
    assert np.allclose(G["home_hand_pose_r"], o2.task.home_hand_pose[:])
    st2, _ = o2.batch_reset(1)
    ref2 = G["pushI_reset_state"]
    assert np.abs(st2[0, :ox] - ref2[:ox]).max() < 1e-12


# ------------------------------------------------------------------ the drop-in env classes (device algorithm through the CPU lane emulation)
from pybullet_robot_envs.envs import iCubReachGymEnv, iCubPushGymEnv, iCubPushGymGoalEnv  # noqa: E402

ENVS = [
    (iCubReachGymEnv, "reach", "reachG", dict(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, max_steps=5)),
    (iCubPushGymEnv, "push", "pushH", dict(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0.0, tg_pose_rnd_std=0, max_steps=1000, reward_type=0)),
    (iCubPushGymEnv, "pushr", "pushI", dict(use_IK=1, control_arm='r', control_orientation=1, max_steps=6, reward_type=1)),
    (iCubPushGymEnv, "pushj", "pushJ", dict(use_IK=0, control_arm='l', max_steps=1000, reward_type=1)),
    (iCubPushGymGoalEnv, "goal", "goalK", dict(use_IK=1, control_arm='r', control_orientation=1, obj_pose_rnd_std=0.0, tg_pose_rnd_std=0, max_steps=4)),
    (iCubReachGymEnv, "reach", "repS", dict(action_repeat=2, use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, max_steps=4)),
]


@pytest.mark.parametrize("cls,stag,tag,kw", ENVS)
def test_env_classes_match_reference(emu_lib, cls, stag, tag, kw):
    replay_env(emu_lib, cls, stag, tag, kw)


def replay_env(lib, cls, stag, tag, kw):
    """Spaces bit-identical to the reference's; step() returns what the reference classes returned (scaled obs, reward,
    done, counter, is_success) step by step from the reference's own states."""
    env = cls(_lib=lib, **kw)
    goal = stag == "goal"
    box = env.observation_space["observation"] if goal else env.observation_space
    assert box.low.dtype == np.float32 and box.low.tobytes() == G[stag + "_obs_low"].tobytes()
    assert box.high.tobytes() == G[stag + "_obs_high"].tobytes()
    assert env.action_space.low.tobytes() == G[stag + "_act_low"].tobytes() and env.action_space.high.tobytes() == G[stag + "_act_high"].tobytes()
    if stag in ("reach", "push"):
        o = env.reset()
        assert np.abs(o - G[tag + "_reset_obs"]).max() < 2e-3
    pre, act = G[tag + "_pre_state"], G[tag + "_actions"]
    for k in range(len(act)):
        env._engine.set_state(pre[k:k + 1].astype(np.float32))
        ob, r, d, info = env.step(act[k])
        ob = ob["observation"] if goal else ob
        assert ob.dtype == np.float64 and ob.shape == G[tag + "_obs"][k].shape
        check_scaled_obs(ob, G[tag + "_obs"][k], tag, k)
        assert abs(float(r) - G[tag + "_reward"][k]) < 2e-5 * max(1, abs(G[tag + "_reward"][k]))
        assert float(d) == G[tag + "_done"][k]
        assert int(env._env_step_counter) == G[tag + "_counter"][k]
        if goal:
            assert bool(info["is_success"]) == bool(G[tag + "_success"][k])
    if stag == "pushr":   # host-side restatements of the reference helpers agree with what step() returned
        assert abs(float(env._compute_reward()) - float(r)) < 1e-4 and float(env._termination()) == float(d)


def test_registered_ids():
    import pybullet_robot_envs
    ids = [i for i, _, _ in pybullet_robot_envs._IDS]
    assert ids[:3] == ['iCubReach-v0', 'iCubPush-v0', 'iCubPushGoal-v0']


def config1_trace(lib, steps=500):
    """BASELINE config 1: iCubReach-v0 kwargs, 1 env, a_t = 0.5 [sin 0.05t, cos 0.05t, sin 0.03t], closed loop."""
    env = iCubReachGymEnv(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, max_steps=1000, _lib=lib)
    obs = [env.reset()]
    rew, done = [], []
    for t in range(steps):
        a = 0.5 * np.array([np.sin(0.05 * t), np.cos(0.05 * t), np.sin(0.03 * t)])
        o, r, d, _ = env.step(a)
        obs.append(o); rew.append(float(r)); done.append(float(d))
    return np.array(obs), np.array(rew), np.array(done)


def check_config1(lib, steps):
    """fp32 device path vs the fp64 trace.  Up to step ~250 the rollout is smooth and the two agree to 1e-3; later the commanded
    hand pose sits on the workspace boundary, the arm works against its joint limits (hand velocities of several m/s in
    the reference trace itself) and the closed loop is chaotic, so only finiteness and the termination flags are compared
    there."""
    obs, rew, done = config1_trace(lib, steps)
    smooth = min(steps, 250)
    ref = G["cfg1_obs_every10"][:smooth // 10 + 1]
    err = np.abs(obs[::10][:len(ref)] - ref)
    assert err.max() < 1e-2, err.max()            # scaled observations, closed loop
    assert np.abs(rew[:smooth] - G["cfg1_reward"][:smooth]).max() < 5e-3
    assert np.isfinite(obs).all() and np.isfinite(rew).all() and (done == G["cfg1_done"][:steps]).all()


def test_config1_trace_prefix(emu_lib):
    check_config1(emu_lib, 60)                     # the CPU lane emulation is slow: the first 60 steps; all 500 on the GPU


def test_icub_model_compiler():
    """model/sdf.py output (committed JSON): tree structure, masses and the base pinned at the fixed constraint's rest pose."""
    from pybullet_robot_envs.model.table import icub_model, icub_info, ICUB_HOME
    full, sim = icub_model(full=True), icub_model()
    assert len(full["links"]) == 38 and sum(1 for l in full["links"] if l["jtype"]) == 32
    assert len(sim["links"]) == 22 and sum(1 for l in sim["links"] if l["jtype"]) == 20        # legs (12 DoF + 4 fixed links) pruned
    assert abs(sum(l["mass"] for l in full["links"]) + full["base"]["mass"] - 33.0617) < 1e-3   # icub_model.sdf total mass
    names = [l["joint_name"] for l in full["links"] if l["jtype"]]
    assert set(names) == set(ICUB_HOME) and names[12:15] == ["torso_pitch", "torso_roll", "torso_yaw"]
    for i, l in enumerate(sim["links"]):
        assert l["parent"] < i                                                                 # topological order survives the pruning
    # base COM raised to 1.2 x its loaded height, orientation R_base^T (icub_env.py:97-103)
    R = np.array(full["base_R"]); p = np.array(full["base_position"]); com = np.array(full["base"]["com"])
    assert abs((p + R @ com)[2] - 1.2 * (0.63 - 0.044081)) < 1e-6
    assert np.allclose(R, [[np.cos(-3.14), -np.sin(-3.14), 0], [np.sin(-3.14), np.cos(-3.14), 0], [0, 0, 1]], atol=1e-12)
    info = icub_info(sim, "r")
    assert [info["dof_names"][d] for d in info["controlled"]][3:] == ["r_shoulder_pitch", "r_shoulder_roll", "r_shoulder_yaw", "r_elbow",
                                                                      "r_wrist_prosup", "r_wrist_pitch", "r_wrist_yaw"]


def test_quaternion_observation_option(emu_lib):
    """control_eu_or_quat=1 on the robot / world classes (icub_env.py:219-224, world_env.py:118-124): orientation entries are the
    quaternion of the same rotation, limits +-1, one more entry."""
    from pybullet_robot_envs import _client
    from pybullet_robot_envs.envs.icub_envs.icub_env import iCubEnv
    from pybullet_robot_envs.envs.world_envs.world_env import WorldEnv, euler_from_quat
    env = iCubReachGymEnv(use_IK=1, _lib=emu_lib)
    env.reset()
    cid = env._physics_client_id
    eu, lim_e = env._robot.get_observation()
    rq = iCubEnv.__new__(iCubEnv); rq.__dict__.update(env._robot.__dict__); rq._control_eu_or_quat = 1
    q, lim_q = rq.get_observation()
    assert len(q) == len(eu) + 1 == rq.get_observation_dim() and lim_q[3:7] == [[-1, 1]] * 4 and rq.get_action_dim() == 3
    assert np.abs(euler_from_quat(np.array(q[3:7])) - np.array(eu[3:6])).max() < 1e-9 and q[7:] == eu[6:]
    w = WorldEnv(cid, control_eu_or_quat=1)
    ow, lw = w.get_observation()
    assert len(ow) == 7 == w.get_observation_dimension() and lw[3:] == [[-1, 1]] * 4 and abs(np.linalg.norm(ow[3:]) - 1) < 1e-6
    env.close()
