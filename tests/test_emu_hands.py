"""iCub with hands (icub_model_with_hands.sdf; 60 simulated DoF, one env per 128-virtual-lane group, persistent motor records,
fingertip contact forces) through the CPU lane emulation of the device algorithm vs the fp64 oracle."""
import numpy as np
import pytest

import parity
from pybullet_robot_envs import _capi


def test_hands_joint_control_and_finger_commands(emu_lib):
    parity.check_hands(_capi.Engine, emu_lib, "r", 0, n=1, steps=3)


def test_hands_ik_control(emu_lib):
    parity.check_hands(_capi.Engine, emu_lib, "l", 1, n=1, steps=2)


def test_hands_fingertip_contacts(emu_lib):
    parity.check_hands_contacts(_capi.Engine, emu_lib, "r")


def test_set_motors_is_hands_only_and_validates(emu_lib, panda):
    eng = _capi.Engine(panda["table"], lib=emu_lib, num_envs=1)
    with pytest.raises(RuntimeError, match="motor record"):
        eng.set_motors([0], [0.0], 0.1)
    eh, ora, info = parity.make_hands_pair(_capi.Engine, emu_lib, 2, "l", 0)
    with pytest.raises(RuntimeError, match="bad DoF"):
        eh.set_motors([60], [0.0], 0.1)
    with pytest.raises(RuntimeError, match="robot-level"):
        parity.make_hands_pair(_capi.Engine, emu_lib, 1, "l", 0, action_repeat=2)
    # a masked command only reaches the selected envs: env 1 closes its hand, env 0 keeps it open
    from pybullet_robot_envs.model.table import GRASP_POS
    st = np.zeros((2, eh.state_floats), np.float32)
    st[:, :60] = np.asarray(info["home"], np.float32)
    st[:, 60:67] = [0.5, -0.03, 0.65, 0, 0, 0, 1]
    st[:, eh.x_off + 5] = 0
    eh.set_state(st)
    eh.set_motors(list(range(60)), info["home"], 0.2)                  # what reset would have written
    eh.set_motors(info["fingers"], GRASP_POS, 0.1, 10.0, mask=[0, 1])
    eh.settle(20)
    s = eh.get_state()
    f = info["fingers"]
    assert np.abs(s[0, f]).max() < 1e-3 and s[1, f[1]] > 0.05


def test_hands_force_limited_reset(emu_lib):
    parity.check_hands_force_limited_reset(_capi.Engine, emu_lib)


def test_hands_apply_action_max_vel(emu_lib):
    """iCubEnv.apply_action(action, max_vel) (icub_env.py:338-360): every commanded motor gets `maxVelocity`; against the oracle."""
    eng, ora, info = parity.make_hands_pair(_capi.Engine, emu_lib, 1, "r", 0)
    eng.reset()
    st, mrec, _ = ora.hands_reset(1)
    s32 = st.astype(np.float32)
    eng.set_state(s32)
    a = (np.asarray(info["home"])[info["controlled"]] + 0.5)[None, :].astype(np.float32)
    eng.apply_action(a, max_vel=0.4)
    so, mrec = ora.hands_apply_action(s32.astype(np.float64), mrec, a, max_vel=0.4)
    mot = eng.get_motor_state()
    assert (mot[0, 3, info["controlled"]] == np.float32(0.4)).all()
    eng.settle(6)
    so = ora.hands_settle(so, mrec, 6)
    se = eng.get_state()
    nd, vo = eng.ndof, eng.v_off
    assert np.abs(se[:, :nd] - so[:, :nd]).max() < 5e-6 and np.abs(se[:, vo:vo + nd] - so[:, vo:vo + nd]).max() < 1e-3
    assert np.abs(so[:, vo:vo + nd]).max() <= 0.4 + 1e-3 and np.abs(so[0, vo + np.array(info["controlled"])]).max() > 0.39     # the bound binds


def test_hands_five_fingertips_on_the_object(emu_lib):
    w = parity.check_hands_five_fingertips(_capi.Engine, emu_lib, "r")
    assert w["fingertips_in_contact"] == 5


def test_helloworld_icub_demo_grasps_and_lifts_the_brick(emu_lib):
    """The reference's scripted grasp (examples/helloworlds/helloworld_icub.py:61-125) on the stand-alone class, one env on the CPU lane
    emulation: the fingers close on the brick (>= 3 fingertips in contact), the brick is lifted with the hand (>= 5 cm; the demo raises
    the hand by 18 cm), carried to the right and dropped back onto the table when the hand opens."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import demo_icub_hands
    from pybullet_robot_envs import _client
    from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv
    cid = _client.connect(1, lib=emu_lib)
    robot = iCubHandsEnv(cid, use_IK=1, control_arm='r')
    obj, tips = [], []

    def log(line):
        obj.append(robot.get_object_pose()[0, :3].copy())
        tips.append(int(np.atleast_1d(robot.check_contact_fingertips()[0])[0]))
    demo_icub_hands.run(robot, log=log)
    # after reset, 1 above, 2 turned, 3 closed, 4 up, 5 right, 6 open
    assert tips[3] >= 3 and tips[4] >= 3, tips
    assert obj[4][2] - obj[2][2] >= 0.05, obj
    assert np.hypot(obj[5][0] - 0.3, obj[5][1] + 0.2) < 0.08 and obj[5][2] > obj[2][2] + 0.05, obj
    # (released 25 cm above the table it tumbles: it comes to rest on one of its faces, 2.5 or 3.75 cm half height)
    assert 0.625 + 0.02 < obj[6][2] < 0.625 + 0.045 and tips[6] == 0, (obj, tips)
    _client.disconnect(cid)


def test_scripted_grasp_against_the_oracle(emu_lib):
    """config 5 closed loop: the reference's grasp demo on the engine and, command by command, on the fp64 oracle -- both close on the
    brick, lift it ~19 cm, carry and release it; the brick's position is compared per phase (parity.check_hands_demo_against_oracle)"""
    rep = parity.check_hands_demo_against_oracle(emu_lib, n=1)
    print("scripted grasp, engine vs oracle:", rep)
    assert rep["lift_oracle_m"] > 0.15 and rep["lift_engine_m"] > 0.15



def test_hands_solver_residual_threshold(emu_lib):
    parity.check_hands_residual_threshold(_capi.Engine, emu_lib, n=1, steps=2)
