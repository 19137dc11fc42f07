"""iCub (32 DoF, one env per 64-lane group) through the CPU lane emulation of the device algorithm vs the fp64 oracle."""
import pytest

import parity
from pybullet_robot_envs import _capi


@pytest.mark.parametrize("task,arm,use_ik,ori,rt", [(0, "l", 1, 0, 1), (1, "r", 1, 1, 1), (1, "l", 0, 0, 0), (2, "r", 1, 1, 1)])
def test_icub_reset_and_steps(emu_lib, task, arm, use_ik, ori, rt):
    parity.check_icub(_capi.Engine, emu_lib, task, arm, use_ik, ori, rt, n=1, steps=3)


def test_icub_auto_reset(emu_lib):
    """PBRE_F_AUTO_RESET in the lane-group core (iCub push, IK control): snapshot reset vs an explicit masked reset."""
    from pybullet_robot_envs.model.table import icub_table
    tbl, model, info = icub_table("l")
    ov = parity.icub_overrides(info, "l", 1, 0, 1)
    parity.check_auto_reset(_capi.Engine, emu_lib, tbl, n=2, max_steps=2, act_dim=3, robot=_capi.ROBOT_ICUB, **ov)
