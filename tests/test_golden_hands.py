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
