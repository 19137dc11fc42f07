"""pandaEnv used alone (robot-level interface: apply_action / pre_grasp / grasp / check_*; reference panda_env.py:195-365 and
examples/helloworlds/helloworld_panda.py) through the CPU lane emulation of the device algorithm vs the fp64 oracle."""
import numpy as np
import pytest

import parity
from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs.envs.panda_envs.panda_env import pandaEnv


@pytest.mark.parametrize("use_ik,ori", [(0, 1), (1, 1), (1, 0)])
def test_panda_arm_commands(emu_lib, use_ik, ori):
    """joint control, IK with (6-D) and without (3-D: the home orientation is kept) orientation control; finger commands with force
    and velocity bounds; apply_action(max_vel)"""
    parity.check_panda_arm(_capi.Engine, emu_lib, use_ik, ori, n=1, steps=3)


def test_panda_arm_grasp_contacts(emu_lib):
    parity.check_panda_arm_grasp(_capi.Engine, emu_lib, n=1, steps=2)


def test_helloworld_panda_demo_lifts_the_object(emu_lib):
    """The reference demo's phases on the stand-alone class: the hand reaches the commanded poses and the closed fingers carry the
    object up with the hand."""
    cid = _client.connect(1, lib=emu_lib)
    robot = pandaEnv(cid, use_IK=1)
    poses = parity.run_panda_demo(robot)
    obs, lim = robot.get_observation()
    assert len(obs) == 18 and len(lim) == 18 and np.abs(np.array(obs[:3]) - [0.5, 0.0, 0.9]).max() < 5e-3
    assert abs(poses[1][0, 2] - 0.65) < 2e-3                    # resting on the table (top at 0.625) before the grasp
    assert poses[3][0, 2] > 0.80                                # lifted with the hand
    n, f = robot.check_contact_fingertips(0)
    assert n == 2 and min(f) > 1.0
    _client.disconnect(cid)


def test_stand_alone_observation_options(emu_lib):
    """control_eu_or_quat=1 (quaternion instead of Euler angles), includeVelObs=False, control_orientation=0 (3-D IK commands)"""
    cid = _client.connect(2, lib=emu_lib)
    robot = pandaEnv(cid, use_IK=1, control_orientation=0, control_eu_or_quat=1, includeVelObs=False)
    obs, lim = robot.get_observation()
    assert obs.shape == (2, 16) and len(lim) == 16 and robot.get_observation_dim() == 16 and robot.get_action_dim() == 3
    assert lim[3:7] == [[-1, 1]] * 4 and abs(np.linalg.norm(obs[0, 3:7]) - 1) < 1e-6
    robot.apply_action([0.4, 0.1, 0.85]); robot.step_simulation(150)
    o2, _ = robot.get_observation()
    assert np.abs(o2[:, :3] - [0.4, 0.1, 0.85]).max() < 5e-3
    assert np.abs(np.abs((o2[:, 3:7] * obs[:, 3:7]).sum(1)) - 1).max() < 1e-3       # the home orientation is kept
    _client.disconnect(cid)
    # the same robot inside a task env: commands belong to the env's step()
    from pybullet_robot_envs.envs import pandaPushGymEnv
    env = pandaPushGymEnv(_lib=emu_lib)
    with pytest.raises(RuntimeError, match="fused step"):
        env._robot.pre_grasp()
    env.close()


def test_scripted_grasp_against_the_oracle(emu_lib):
    """the reference's helloworld_panda.py grasp on the engine and, command by command, on the fp64 oracle: both lift the object 22 cm;
    its position after the lift agrees within 5 mm (measured 0.2 mm; parity.check_panda_demo_against_oracle)"""
    rep = parity.check_panda_demo_against_oracle(emu_lib, n=1)
    print("Panda scripted grasp, engine vs oracle:", rep)
    assert rep["lift_oracle_m"] > 0.2 and rep["lift_engine_m"] > 0.2


def test_closed_fingers_press_with_exactly_their_force_limit(emu_lib):
    """analytic KAT (parity.check_finger_force_kat): the contact forces on a force-limited finger add up to its 10 N bound"""
    print("finger forces:", parity.check_finger_force_kat(emu_lib, n=1))

