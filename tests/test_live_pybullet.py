"""Optional live cross-check against real PyBullet (skips with "pybullet absent" where it is not installed -- which is every box
this build has seen; see tests/live_pybullet.py).  The URDF writer is checked on CPU regardless: the text it emits parses back, with
the engine's own URDF reader, to the RobotTable the engine steps."""
import os
import tempfile

import numpy as np
import pytest

import live_pybullet
from pybullet_robot_envs.model import urdf as urdf_reader
from pybullet_robot_envs.model.table import panda_table, build_table, PANDA_SPHERES


def test_urdf_writer_round_trips_to_the_same_robot_table():
    tbl, model = panda_table()
    text = live_pybullet.model_to_urdf(model, PANDA_SPHERES)
    fd, path = tempfile.mkstemp(suffix=".urdf")
    os.write(fd, text.encode()); os.close(fd)
    try:
        m2 = urdf_reader.parse_urdf(path, base_position=model["base_position"])
    finally:
        os.remove(path)
    ee = int(tbl[4])
    t2 = build_table(m2, PANDA_SPHERES, ee_link=ee)
    assert t2.shape == tbl.shape and np.abs(t2 - tbl).max() < 1e-12


def test_engine_matches_live_pybullet(emu_lib):
    """Free-space joint trajectories and the settled cube of the engine (lane emulation here, HIP on the GPU box) against PyBullet
    stepping the same model; tolerances: joint angles 1e-4 rad over 50 steps, cube rest height 1 mm."""
    pytest.importorskip("pybullet", reason="pybullet absent: physics parity stays unpinned (DESIGN.md section 2)")
    from pybullet_robot_envs import _capi
    tbl, _ = panda_table()
    live = live_pybullet.LivePandaPush(seed=0, obj_pose_rnd_std=0.0, tg_pose_rnd_std=0.0)
    q, qd, obj, tw = live.reset()
    eng = _capi.Engine(tbl, task=1, num_envs=1, lib=emu_lib, obj_pose_rnd_std=0.0, tg_pose_rnd_std=0.0)
    eng.reset()
    st = eng.get_state()
    assert abs(st[0, 11] - obj[2]) < 1e-3                           # K3: settled cube height
    st[0, :9] = q; st[0, 16:25] = qd; st[0, 9:16] = obj; st[0, 25:31] = tw
    eng.set_state(st)
    rng = np.random.default_rng(0)
    for k in range(50):
        a = rng.uniform(-1, 1, 7)
        (q, qd, obj, tw), r, d = live.step(a)
        eng.step(a[None].astype(np.float32))
    se = eng.get_state()
    assert np.abs(se[0, :9] - q).max() < 1e-4 and np.abs(se[0, 9:12] - obj[:3]).max() < 1e-3
    live.close()
