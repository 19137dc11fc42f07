"""pandaEnv used alone (robot-level interface) on the GPU: the half-wave motor-record engine (pbre_wide.hip, ShapePA) through the
C-ABI against the fp64 oracle, and the reference's helloworld_panda.py demo on a batch."""
import numpy as np
import pytest

import parity
from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs.envs.panda_envs.panda_env import pandaEnv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_ik,ori", [(0, 1), (1, 1), (1, 0)])
def test_panda_arm_commands(hip_lib, use_ik, ori):
    eng = parity.check_panda_arm(_capi.Engine, hip_lib, use_ik, ori, n=5, steps=4)
    assert eng.kernel_info()[4] == 5


def test_panda_arm_grasp_contacts(hip_lib):
    parity.check_panda_arm_grasp(_capi.Engine, hip_lib, n=3, steps=3)


def test_helloworld_panda_demo_on_a_batch(hip_lib):
    """2048 replicas of the demo: every env reaches the poses, closes the fingers on the object and lifts it; replicas identical."""
    n = 2048
    cid = _client.connect(n, lib=hip_lib)
    robot = pandaEnv(cid, use_IK=1)
    poses = parity.run_panda_demo(robot)
    obs, lim = robot.get_observation()
    assert obs.shape == (n, 18) and np.abs(obs[:, :3] - [0.5, 0.0, 0.9]).max() < 5e-3
    assert (poses[3][:, 2] > 0.80).all()
    nt, f = robot.check_contact_fingertips(0)
    assert (nt == 2).all() and f.min() > 1.0
    st = robot._client.engine.get_state()
    assert np.isfinite(st).all() and np.array_equal(st, np.broadcast_to(st[0], st.shape))
    _client.disconnect(cid)


def test_scripted_grasp_against_the_oracle(hip_lib):
    """the reference's helloworld_panda.py grasp on the engine and, command by command, on the fp64 oracle: both lift the object 22 cm;
    its position after the lift agrees within 5 mm (measured 0.2 mm; parity.check_panda_demo_against_oracle)"""
    rep = parity.check_panda_demo_against_oracle(hip_lib, n=2)
    print("Panda scripted grasp, engine vs oracle:", rep)
    assert rep["lift_oracle_m"] > 0.2 and rep["lift_engine_m"] > 0.2


def test_closed_fingers_press_with_exactly_their_force_limit(hip_lib):
    """analytic KAT (parity.check_finger_force_kat): the contact forces on a force-limited finger add up to its 10 N bound"""
    print("finger forces:", parity.check_finger_force_kat(hip_lib, n=4))

