"""iCub golden vectors captured from the reference's own classes (tools/make_golden.py: iCubReachGymEnv,
iCubPushGymEnv, iCubPushGymGoalEnv executed over the oracle's physics) pin the iCub glue restated in oracle/:
observation order (COM-frame hand pose, raw velocity, the 10 controlled joints), the IK-mode hand-pose accumulation
(scales 0.005 / 0.01 / 0.02, Euler and workspace clipping per arm), joint-mode targets, the three reward variants,
termination and counter logic.  CPU only."""
import os

import numpy as np
import pytest

import orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "icub_glue.npz"))

#        tag      arm task use_ik ori max_steps reward_type
CASES = [("reachG", "l", 0, 1, 0, 5, 1), ("pushH", "l", 1, 1, 0, 1000, 0), ("pushI", "r", 1, 1, 1, 6, 1),
         ("pushJ", "l", 1, 0, 0, 1000, 1), ("goalK", "r", 2, 1, 1, 4, 1), ("goalL", "r", 2, 1, 1, 4, 1)]


def make(arm, task, use_ik, ori, max_steps, reward_type):
    o, tbl, info = orc.icub_oracle(arm, task=task, use_ik=use_ik, control_orientation=ori)
    o.task.max_steps = max_steps
    o.task.reward_type = reward_type
    return o, info


@pytest.mark.parametrize("tag,arm,task,use_ik,ori,max_steps,reward_type", CASES)
def test_oracle_glue_reproduces_reference(tag, arm, task, use_ik, ori, max_steps, reward_type):
    o, info = make(arm, task, use_ik, ori, max_steps, reward_type)
    pre, act = G[tag + "_pre_state"], G[tag + "_actions"]
    ox = o.state_floats - 16
    assert pre.shape[1] == o.state_floats == 144 and act.shape[1] == o.task.n_act
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


def test_reset_and_bookkeeping():
    o, info = make("l", 0, 1, 0, 5, 1)
    st, obs = o.batch_reset(1)
    ref = G["reachG_reset_state"]
    ox = 128
    assert np.abs(st[0, :ox] - ref[:ox]).max() < 1e-12                 # IK at reset + 1 + 100 + 101 steps, settled object
    assert np.abs(st[0, ox + 6:ox + 12] - ref[ox + 6:ox + 12]).max() < 1e-12
    # joint bookkeeping the reference derives by name (icub_env.py:107-150)
    names = info["dof_names"]
    link_of = {n: i for i, n in enumerate([l for l in range(38)])}
    from pybullet_robot_envs.model.table import icub_table
    _, model, info_r = icub_table("r")
    jidx = [i for i, l in enumerate(model["links"]) if l["jtype"] != 0]
    assert [jidx[d] for d in info["controlled"]] == list(G["joints_to_control_l"])
    assert [jidx[d] for d in info_r["controlled"]] == list(G["joints_to_control_r"])
    assert [info["ee_link"], info_r["ee_link"]] == list(G["end_eff_idx"])
    assert np.allclose(G["home_hand_pose_l"], o.task.home_hand_pose[:]) 
    o2, _ = make("r", 1, 1, 1, 6, 1)
    assert np.allclose(G["home_hand_pose_r"], o2.task.home_hand_pose[:])
    st2, _ = o2.batch_reset(1)
    ref2 = G["pushI_reset_state"]
    assert np.abs(st2[0, :ox] - ref2[:ox]).max() < 1e-12
