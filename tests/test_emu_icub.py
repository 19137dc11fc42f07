"""iCub (32 DoF, one env per 64-lane group) through the CPU lane emulation of the device algorithm vs the fp64 oracle."""
import numpy as np
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


def test_icub_action_repeat(emu_lib):
    n = 2
    eng, ora, info = parity.make_icub_pair(_capi.Engine, emu_lib, n, task=0, control_arm="l", use_ik=1, control_orientation=0,
                                           action_repeat=2, max_steps=4)
    eng.reset()
    st_o, _ = ora.batch_reset(n)
    rng = np.random.default_rng(5)
    xo = eng.x_off
    for k in range(5):
        a = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        ob, rw, dn = eng.step(a)
        st_o, out = ora.batch_step(st_o, a)
        se = eng.get_state()
        assert np.array_equal(se[:, xo + 3], st_o[:, xo + 3]) and (dn == out[:, -1]).all()
        assert np.abs(se[:, xo + 6:xo + 12] - st_o[:, xo + 6:xo + 12]).max() < 1e-6
        assert parity.rel(ob, out[:, :-2]).max() < 2e-2
    assert list(se[:, xo + 3]) == [5, 5]


def test_implicit_joint_damping_option(emu_lib, panda):
    parity.check_implicit_damping(_capi.Engine, emu_lib, panda["table"], n=2)


def test_icub_implicit_joint_damping(emu_lib):
    eng, ora, info = parity.make_icub_pair(_capi.Engine, emu_lib, 1, task=1, control_arm="l", use_ik=0, phys={"implicit_joint_damping": 1})
    ora.params.implicit_joint_damping = 1
    eng.reset()
    st, _ = ora.batch_reset(1)
    xo = eng.x_off
    assert parity.rel(eng.get_state()[:, :xo], st[:, :xo]).max() < 2e-3
    a = np.random.default_rng(2).uniform(-1, 1, (1, eng.act_dim)).astype(np.float32)
    eng.set_state(st.astype(np.float32))
    ob, rw, dn = eng.step(a)
    so, out = ora.batch_step(st.astype(np.float32).astype(np.float64), a)
    assert parity.rel(eng.get_state()[:, :xo], so[:, :xo]).max() < 2e-3 and parity.rel(ob, out[:, :-2]).max() < 2e-2


def test_icub_force_limited_motors(emu_lib):
    parity.check_icub_force_limited(_capi.Engine, emu_lib)


def test_icub_object_rows_split(emu_lib):
    parity.check_obj_split(_capi.Engine, emu_lib, n=2)


def test_icub_full_model_one_env_per_64_lanes(emu_lib):
    parity.check_icub_full_model(_capi.Engine, emu_lib, n=1, steps=2)


def test_icub_neighbour_independence(emu_lib):
    parity.check_wave_neighbour_independence(_capi.Engine, emu_lib, steps=2)


def test_icub_full_episode_rollout(emu_lib):
    """2000 free-running steps (a whole iCub episode, joint control) against the oracle with stated drift bounds"""
    parity.check_icub_full_episode(_capi.Engine, emu_lib, n=2, steps=2000)


def test_icub_reset_snapshot(emu_lib):
    """pbre_reset_snapshot on the lane-group engine (iCub push: incl. the initial distances of the normalised reward)"""
    from pybullet_robot_envs.model.table import icub_table
    tbl, model, info = icub_table("l")
    ov = parity.icub_overrides(info, "l", 1, 0, 1)
    parity.check_reset_snapshot(_capi.Engine, emu_lib, tbl, n=4, robot=_capi.ROBOT_ICUB, **ov)


def test_icub_lane_path_matches_lane_group_kernel(emu_lib, monkeypatch):
    """CPU emulation: the lane-per-env step (pbre_lane.hpp, Lane::step / finish / ik_targets) against the lane-group core"""
    assert parity.check_icub_lane_ab(_capi.Engine, emu_lib, monkeypatch, 1, n=3, steps=8) < 2e-3


def test_icub_hand_on_table(emu_lib):
    print(parity.check_icub_table_contact(_capi.Engine, emu_lib, n=2, steps=45))


# ---------------------------------------------------------------------------------------------- iCubEnv used alone
@pytest.mark.parametrize("arm,use_ik,ori", [("l", 0, 1), ("l", 1, 1), ("r", 1, 0)])
def test_icub_env_robot_level_commands(emu_lib, arm, use_ik, ori):
    """iCubEnv.apply_action(action, max_vel) + stepSimulation loops on the stand-alone class (reference icub_env.py:91-151, 259-360):
    joint control, IK with (6-D) and without (3-D: the home orientation is kept) orientation control, either arm"""
    parity.check_icub_arm(_capi.Engine, emu_lib, arm, use_ik, ori, n=1, steps=3)


def test_icub_env_inside_a_task_env_refuses_robot_level_commands(emu_lib):
    from pybullet_robot_envs.envs import iCubReachGymEnv
    env = iCubReachGymEnv(use_IK=1, num_envs=1, _lib=emu_lib)
    with pytest.raises(RuntimeError):
        env._robot.apply_action([0.3, 0.2, 0.8])
    env.close()


def test_icub_crafted_contact_states(emu_lib):
    """hand on the object / on the table / both / a joint beyond its limit: one step each against the oracle, per quantity"""
    rep = parity.check_icub_contact_states(_capi.Engine, emu_lib, n_each=3)
    assert rep["states"] == 12


@pytest.mark.parametrize("obj_name", ["YcbTennisBall", "duck_vhacd"])
def test_icub_crafted_contact_states_round_objects(emu_lib, monkeypatch, obj_name):
    """the same with a round object (sphere / upright cylinder primitive), through the lane-per-env code (PBRE_ICUB_LANE=1: the device
    pipeline's dynamics and classifier; complex envs by the lane-group code) -- iCubReach-v0's default object is the duck"""
    monkeypatch.setenv("PBRE_ICUB_LANE", "1")
    rep = parity.check_icub_contact_states(_capi.Engine, emu_lib, n_each=2, obj_name=obj_name)
    assert rep["states"] == 8


def test_icub_push_closed_loop_against_oracle(emu_lib):
    rep = parity.check_icub_push_closed_loop(_capi.Engine, emu_lib, n=2)
    assert rep["touched_envs"] == 2


def test_icub_nan_inf_guard(emu_lib):
    """NaN / Inf guard through Lane::step / Lane::finish and (complex envs, masked paths) Core::step / Core::observe on the lane emulation"""
    parity.check_icub_nan_guard(_capi.Engine, emu_lib, n=8)


@pytest.mark.parametrize("task", [0, 1])
def test_icub_solver_residual_threshold(emu_lib, task):
    """pbre_physics.solver_residual_threshold on the iCub's lane-group kernel (Core::step<RT>, half-wave shape): joint-control steps against
    the oracle with the same threshold, equal per-env sweep counts."""
    eng, ora, info = parity.make_icub_pair(_capi.Engine, emu_lib, 4, task, "l", 0, 0, obj_std=0.05, tg_std=0.2)
    eng.reset(); st, _ = ora.batch_reset(4)
    parity.check_group_residual_threshold(eng, ora, st, np.random.default_rng(5), parity.TOL_ICUB, steps=3)


@pytest.mark.parametrize("use_ik", [0, 1])
def test_icub_floating_base_option(emu_lib, use_ik):
    """Fidelity option for the reference's floating base held by createConstraint(JOINT_FIXED) (icub_env.py:95-101): the base as a dynamic
    body -- six virtual joints held by the constraint's equivalent motors, legs lumped (model/table.py: float_base) -- on the 64-lane
    shape: against the oracle on the same model, and its (small, measured) effect against the rigidly pinned default."""
    rep = parity.check_icub_floating_base(_capi.Engine, emu_lib, n=2, steps=3, use_ik=use_ik)
    assert rep["ee_pos_shift_vs_pinned_base_40_steps_m"] < 3e-2      # (12 mm with the constraint's 500 N bound on its rows, round 6; 2 mm unbounded)


def test_icub_base_constraint_force_bound(emu_lib):
    """the 500 N of the base constraint as a bound on its rows (Tables::mforce / orc_model.max_force), exercised with a bound below the robot's
    weight: the vertical row saturates, the robot sinks at the free-fall deficit, engine and oracle agree"""
    rep = parity.check_icub_base_force_bound(_capi.Engine, emu_lib)
    print(rep)


def test_icub_floating_base_through_the_gym_classes(emu_lib):
    from pybullet_robot_envs.envs import iCubReachGymEnv
    a = iCubReachGymEnv(num_envs=2, _lib=emu_lib, obj_pose_rnd_std=0.05)
    b = iCubReachGymEnv(num_envs=2, _lib=emu_lib, obj_pose_rnd_std=0.05, floating_base=True)
    oa, ob = a.reset(), b.reset()
    assert oa.shape == ob.shape == (2, 31) and b._engine.ndof == 26 and b._engine.obj_off == 32
    assert np.abs(oa - ob).max() < 1e-3
    rng = np.random.default_rng(0)
    for _ in range(10):
        act = rng.uniform(-1, 1, (2, a.action_space.shape[0]))
        (oa, ra, da, _), (ob, rb, db, _) = a.step(act), b.step(act)
    assert np.abs(oa - ob).max() < 2e-2 and (da == db).all()


def _write_chamfered_box_obj(path, full=(0.09, 0.07, 0.08), b=0.012):
    """a synthetic "duck_vhacd.obj": a box of the duck stand-in's extents with its corners cut off (24 hull vertices) plus a few interior
    vertices a real mesh would have -- no mesh of the reference's objects exists on any box"""
    hx, hy, hz = (0.5 * x for x in full)
    vs = []
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                vs += [(sx * (hx - b), sy * hy, sz * hz), (sx * hx, sy * (hy - b), sz * hz), (sx * hx, sy * hy, sz * (hz - b))]
    vs += [(0.0, 0.0, 0.0), (0.01, -0.01, 0.02), (-0.02, 0.01, -0.01)]
    with open(path, "w") as f:
        f.write("# synthetic test mesh\n")
        for v in vs:
            f.write("v %.6f %.6f %.6f\n" % (v[0] + 0.003, v[1] - 0.002, v[2] + 0.001))      # (mesh origin off the centre of mass)
        f.write("f 1 2 3\n")


def test_icub_hull_object_from_a_mesh_file(emu_lib, tmp_path, monkeypatch):
    """SURVEY 8(f4) end to end: `obj_name` -> mesh file (PBRE_OBJECT_MESH_DIR; pybullet_data / pybullet_object_models where importable) ->
    convex hull (model/objects.py: hull_physics: <= 32 vertices about the centre of mass, mass properties of the solid) ->
    pbre_set_object_hull -> the iCub engine's lane-group kernel (W = 32: one candidate pass), crafted contact states against the oracle"""
    from pybullet_robot_envs.model import objects
    _write_chamfered_box_obj(str(tmp_path / "duck_vhacd.obj"))
    assert objects.object_physics("duck_vhacd")["obj_shape"] == 2            # no mesh anywhere: the cylinder stand-in
    monkeypatch.setenv("PBRE_OBJECT_MESH_DIR", str(tmp_path))
    ph = objects.object_physics("duck_vhacd")
    assert ph["obj_shape"] == 3 and len(ph["obj_hull"]) == 24 and abs(ph["obj_h"][0] - 0.045) < 1e-9 and abs(ph["obj_h"][2] - 0.04) < 1e-9
    assert np.abs(np.asarray(ph["obj_hull"]).mean(0)).max() < 1e-9           # about the centre of mass (a symmetric solid: its centroid)
    rep = parity.check_icub_contact_states(_capi.Engine, emu_lib, n_each=2, obj_name="duck_vhacd")
    assert rep["states"] == 4 * 2 and rep["complex_envs_stepped"] >= 0
