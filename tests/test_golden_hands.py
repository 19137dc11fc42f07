"""iCubHandsEnv (batched) against vectors captured from the REFERENCE class icub_env_with_hands.iCubHandsEnv run over stub
pybullet (tools/make_golden_hands.py -> tests/golden/icub_hands_glue.npz): joint bookkeeping incl. the `a or b and c` joint
selection, joint ranges, the motor commands of open_hand / pre_grasp / grasp / apply_action, fingertip contact statistics."""
import os

import numpy as np
import pytest

from pybullet_robot_envs import _client
from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icub_hands_glue.npz"))


@pytest.fixture(scope="module", params=["l", "r"])
def robot(request, emu_lib):
    cid = _client.connect(1, lib=emu_lib)
    r = iCubHandsEnv(cid, use_IK=0, control_arm=request.param)
    yield r, "hands_%s_" % request.param
    _client.disconnect(cid)


def test_joint_bookkeeping(robot):
    r, t = robot
    names = list(r._joint_name_to_ids.keys())
    assert names == list(G[t + "joint_names"])
    assert [r._joint_name_to_ids[n] for n in names] == list(G[t + "joint_ids"])
    assert r._joints_to_control == list(G[t + "joints_to_control"]) and len(r._joints_to_control) == 37
    assert r._joints_to_block == list(G[t + "joints_to_block"])
    assert r.end_eff_idx == int(G[t + "end_eff_idx"])
    for k, v in zip(("ll", "ul", "jr", "rs", "jd"), (r.ll, r.ul, r.jr, r.rs, r.jd)):
        assert np.allclose(v, G[t + k], atol=1e-12), k
    assert np.allclose(r._home_hand_pose, G[t + "home_hand_pose"]) and np.allclose(r._eu_lim, G[t + "eu_lim"])
    assert np.allclose(r._workspace_lim, G[t + "workspace"]) and np.allclose(r._com_to_link_hand_frame()[0], G[t + "com_T_link"])
    assert r.get_action_dim() == int(G[t + "action_dim"])
    assert r.fingertip_indices() == list(G[t + "tips"])


def _motor_rows(r, gold):
    """golden rows (joint index, target, positionGain, velocityGain, force or -1) -> {dof: (target, kp, force scale)}"""
    sim = r._info["dof_names"]
    out = {}
    for j, tg, kp, kd, f in gold:
        name = r._model["links"][int(j)]["joint_name"]
        assert kd == 1.0
        if name in sim:                        # joints of the pruned legs have no motor in the engine
            out[sim.index(name)] = (tg, kp, 1.0 if f < 0 else f * (1.0 / 240.0) / (100000.0 / 240.0))
    return out


def test_motor_commands(robot):
    r, t = robot
    eng = r._engine
    m0 = eng.get_motor_state()[0]
    want = _motor_rows(r, G[t + "reset_motors"])              # reset: every joint at its initial position, gain 0.2
    assert len(want) == 60
    for d, (tg, kp, fs) in want.items():
        assert abs(m0[0, d] - tg) < 1e-6 and abs(m0[1, d] - kp) < 1e-7 and m0[2, d] == fs
    pos_cl = [0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 1.57, 0.8, 0.5, 0.8]
    for tag, call in (("open_hand", r.open_hand), ("pre_grasp", r.pre_grasp), ("grasp", r.grasp), ("grasp_pos", lambda: r.grasp(pos_cl))):
        before = eng.get_motor_state()[0]
        call()
        after = eng.get_motor_state()[0]
        want = _motor_rows(r, G[t + tag])
        assert len(want) == 20
        for d in range(60):
            if d in want:
                tg, kp, fs = want[d]
                assert abs(after[0, d] - tg) < 1e-6 and abs(after[1, d] - kp) < 1e-7 and abs(after[2, d] - fs) < 1e-9, (tag, d)
            else:
                assert (after[:, d] == before[:, d]).all(), (tag, d)
    # joint control: one absolute target per controlled joint, clipped to the limits, gain 0.5, default force
    act = G[t + "apply_action_in"]
    before = eng.get_motor_state()[0]
    r.apply_action(list(act))
    after = eng.get_motor_state()[0]
    want = _motor_rows(r, G[t + "apply_action"])
    assert len(want) == 37
    for d in range(60):
        if d in want:
            tg, kp, fs = want[d]
            assert abs(after[0, d] - tg) < 1e-6 and abs(after[1, d] - kp) < 1e-7 and after[2, d] == 1.0, d
        else:
            assert (after[:, d] == before[:, d]).all(), d
    with pytest.raises(AssertionError):
        r.apply_action([0.0] * 10)


def test_fingertip_statistics(robot):
    """check_contact_fingertips / check_collision over the synthetic contact lists the reference class was given: the engine
    reports per-tip mean forces, tips in contact and the number of robot-object contact points in the state record."""
    r, t = robot
    eng = r._engine
    tips = list(G[t + "tips"])
    nd = eng.ndof
    for ci in range(6):
        cp = G[t + "contacts%d_in" % ci]
        f = np.zeros(5); cnt = np.zeros(5)
        for link, force in cp:
            if int(link) in tips:
                f[tips.index(int(link))] += force; cnt[tips.index(int(link))] += 1
        st = eng.get_state()
        st[0, nd + 7:nd + 12] = np.where(cnt > 0, f / np.maximum(cnt, 1), 0.0)
        st[0, nd + 12] = (cnt > 0).sum()
        st[0, nd + 13] = len(cp)
        eng.set_state(st)
        n, forces = r.check_contact_fingertips()
        assert n == int(G[t + "contacts%d_n" % ci])
        assert np.allclose(forces, G[t + "contacts%d_f" % ci], atol=1e-6)
        assert r.check_collision() == bool(G[t + "contacts%d_collision" % ci])


def test_hand_pose_commands_3_6_7_values(emu_lib):
    """IK control: a quaternion command (7 values, used as given) and the Euler command (6 values) of the same pose give the same motor
    targets; with control_orientation=0 the engine takes 3 values and keeps the home orientation; the hand goes where it is told."""
    import math as m
    cid = _client.connect(1, lib=emu_lib)
    r = iCubHandsEnv(cid, use_IK=1, control_arm='r')
    eu = [0.1, 0.2, 1.2]                                    # inside the right arm's Euler limits
    cr, sr, cp, sp, cy, sy = m.cos(eu[0] / 2), m.sin(eu[0] / 2), m.cos(eu[1] / 2), m.sin(eu[1] / 2), m.cos(eu[2] / 2), m.sin(eu[2] / 2)
    q = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
    st = r._engine.get_state()
    r.apply_action([0.3, -0.15, 0.8] + eu)
    m6 = r._engine.get_motor_state()
    r._engine.set_state(st)
    r.apply_action([0.3, -0.15, 0.8] + q)
    m7 = r._engine.get_motor_state()
    assert np.abs(m6 - m7).max() < 2e-4
    assert (m6[0, 1, :60] == np.float32(0.2)).all() and (m6[0, 2, :60] == 1.0).all()     # every joint commanded, gain 0.2, default force
    # Euler commands are clipped to the arm's limits (yaw of the right arm: [0, pi]), quaternions are not
    r._engine.set_state(st); r.apply_action([0.3, -0.15, 0.8, 0.0, 0.0, -1.0]); a = r._engine.get_motor_state()
    r._engine.set_state(st); r.apply_action([0.3, -0.15, 0.8, 0.0, 0.0, 0.0]); b = r._engine.get_motor_state()
    assert np.abs(a - b).max() < 1e-6
    r.step_simulation(40)
    obs, _ = r.get_observation()
    assert np.abs(np.asarray(obs[:3]) - [0.3, -0.15, 0.8]).max() < 5e-3
    with pytest.raises(AssertionError):
        r.apply_action([0.3, 0.0])
    _client.disconnect(cid)
    cid = _client.connect(1, lib=emu_lib)
    r3 = iCubHandsEnv(cid, use_IK=1, control_arm='l', control_orientation=0)
    assert r3.get_action_dim() == 3 and r3._engine.act_dim == 3
    r3.apply_action([0.3, 0.2, 0.85]); r3.step_simulation(40)
    obs, _ = r3.get_observation()
    assert np.abs(np.asarray(obs[:3]) - [0.3, 0.2, 0.85]).max() < 5e-3
    _client.disconnect(cid)
