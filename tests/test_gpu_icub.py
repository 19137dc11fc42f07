"""iCub on the GPU: the 64-lane engine (one env per wavefront, pbre_wide.hip) through the C-ABI against the fp64 oracle."""
import numpy as np
import pytest

import parity
from pybullet_robot_envs import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task,arm,use_ik,ori,rt", [(0, "l", 1, 0, 1), (1, "r", 1, 1, 1), (1, "l", 0, 0, 0), (2, "r", 1, 1, 1), (1, "l", 1, 0, 1)])
def test_icub_reset_and_steps(hip_lib, task, arm, use_ik, ori, rt):
    eng = parity.check_icub(_capi.Engine, hip_lib, task, arm, use_ik, ori, rt, n=6, steps=4)
    info = eng.kernel_info()
    # the lane-per-env pipeline (pbre_lane.hip) steps the batch: [2] flag, [3] simple + [5] complex envs; [1] VGPRs of the lane-group kernel
    assert info[2] == 1 and info[3] + info[5] == 6 and info[4] == 0 and 0 < info[1] <= 256


def test_icub_masked_reset_and_rollout(hip_lib):
    """Free-running rollout (trajectory-level agreement) and a masked reset that leaves the other envs untouched."""
    n = 8
    eng, ora, info = parity.make_icub_pair(_capi.Engine, hip_lib, n, task=1, control_arm="l", use_ik=1, control_orientation=0, obj_std=0.05, tg_std=0.2)
    eng.reset()
    st_o, _ = ora.batch_reset(n)
    rng = np.random.default_rng(0)
    for k in range(40):
        a = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        a[:, 2] = -0.8                                   # push the hand down towards the table / object
        ob, rw, dn = eng.step(a)
        st_o, out = ora.batch_step(st_o, a)
    se = eng.get_state()
    assert np.isfinite(se).all() and np.isfinite(ob).all()
    nd, xo = eng.ndof, eng.x_off
    assert np.abs(se[:, :nd] - st_o[:, :nd]).max() < 2e-2              # joint angles after 40 closed-loop steps
    assert np.abs(se[:, nd:nd + 3] - st_o[:, nd:nd + 3]).max() < 2e-2  # object position
    mask = np.zeros(n, np.uint8); mask[[1, 5]] = 1
    eng.reset(mask)
    s2 = eng.get_state()
    keep = [i for i in range(n) if not mask[i]]
    assert np.array_equal(s2[keep], se[keep])
    assert (s2[[1, 5], xo + 5] == 1).all() and (s2[keep, xo + 5] == 0).all()      # episode numbers
    assert np.abs(s2[[1, 5], xo + 6:xo + 9] - [0.3, 0.26, 0.8]).max() < 1e-6


import test_golden_icub as tgi  # noqa: E402


@pytest.mark.parametrize("cls,stag,tag,kw", tgi.ENVS)
def test_env_classes_match_reference_outputs(hip_lib, cls, stag, tag, kw):
    """The drop-in iCub Gym classes on the GPU against the outputs captured from the reference's own classes."""
    tgi.replay_env(hip_lib, cls, stag, tag, kw)


def test_config1_icub_reach_trace(hip_lib):
    """BASELINE config 1 (iCubReach-v0, 1 env, fixed action sequence, 500 closed-loop steps) against the trace captured from
    the reference class."""
    tgi.check_config1(hip_lib, 500)


def test_icub_full_episode_rollout(hip_lib):
    """A whole 2000-step iCub-push episode (joint control), free running, against the oracle; drift bounds in parity.py."""
    print("iCub full-episode drift:", parity.check_icub_full_episode(_capi.Engine, hip_lib, n=6, steps=2000))


@pytest.mark.parametrize("n", [1, 3, 5])
def test_icub_ragged_batch_sizes(hip_lib, n):
    parity.check_icub(_capi.Engine, hip_lib, 1, "l", 1, 0, 1, n=n, steps=2)


def test_icub_auto_reset(hip_lib):
    from pybullet_robot_envs.model.table import icub_table
    tbl, model, info = icub_table("l")
    ov = parity.icub_overrides(info, "l", 1, 0, 1)
    parity.check_auto_reset(_capi.Engine, hip_lib, tbl, n=40, max_steps=3, act_dim=3, robot=_capi.ROBOT_ICUB, **ov)


def test_icub_action_repeat(hip_lib):
    import test_emu_icub
    test_emu_icub.test_icub_action_repeat(hip_lib)


def test_icub_force_limited_motors(hip_lib):
    parity.check_icub_force_limited(_capi.Engine, hip_lib, n=5, steps=4)


def test_icub_object_rows_split(hip_lib, monkeypatch):
    """the lane-group kernel's object split (PBRE_ICUB_LANE=0: every step by kw_step), bitwise"""
    monkeypatch.setenv("PBRE_ICUB_LANE", "0")
    parity.check_obj_split(_capi.Engine, hip_lib, n=7, steps=4)


def test_icub_lane_pipeline_with_robot_contacts(hip_lib):
    """the lane-per-env pipeline (kw_dyn / kw_quad / kw_fin + kw_list for the envs whose hand touches the object) against the
    lane-group kernel's coupled solve and the oracle"""
    two = parity.check_obj_split(_capi.Engine, hip_lib, n=7, steps=4, exact=False)
    info = two.kernel_info()
    assert info[2] == 1 and info[5] >= 1, info          # some envs were in the complex class (robot contact) at the end


def test_icub_lane_pipeline_matches_lane_group_kernel(hip_lib, monkeypatch):
    """free-running batches: the lane-per-env pipeline against the lane-group kernel, rounding level"""
    print(parity.check_icub_lane_ab(_capi.Engine, hip_lib, monkeypatch, 1, n=96, steps=12))


def test_icub_push_policy_coupled_solve(hip_lib, monkeypatch):
    """the pipeline's coupled robot-object solve (kw_quad_rc) with a third of the batch in contact, against the lane-group kernel at the
    level of the batch"""
    print(parity.check_icub_push_policy(_capi.Engine, hip_lib, monkeypatch))


def test_icub_ik_overlap_is_bit_identical(hip_lib, monkeypatch):
    """Round 5: under Cartesian control the pipeline's solve kernels wait per env for the IK targets (kw_lane_ik publishes an env's targets
    in the iteration in which it converges, on a stream of its own) instead of for the whole IK kernel.  Same arithmetic: 2048 iCub-push envs,
    random hand actions (out-of-reach targets among them: the IK's stragglers), auto-reset -- rows and states bit for bit those of the
    kernel-level dependency (PBRE_IK_OVERLAP=0)."""
    from pybullet_robot_envs.model.table import icub_table
    monkeypatch.setenv("PBRE_ICUB_LANE", "1")
    tbl, model, info = icub_table("l")
    ov = parity.icub_overrides(info, "l", 1, 0, 1)
    n = 2048
    kw = dict(task=_capi.TASK_PUSH, num_envs=n, lib=hip_lib, robot=_capi.ROBOT_ICUB, flags=_capi.F_AUTO_RESET, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2,
              max_steps=30, **ov)
    monkeypatch.setenv("PBRE_IK_OVERLAP", "1")
    a = _capi.Engine(tbl, **kw)
    monkeypatch.setenv("PBRE_IK_OVERLAP", "0")
    b = _capi.Engine(tbl, **kw)
    assert a.kernel_info()[2] == 1 and b.kernel_info()[2] == 1
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(17)
    for k in range(50):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), "step %d" % k
    assert np.array_equal(a.get_state(), b.get_state())
    assert a.kernel_info()[12] == 0 and b.kernel_info()[12] == 0          # (a wait that ran into its bound would have been counted here)


def test_icub_default_path_by_batch_size(hip_lib, monkeypatch):
    """without PBRE_ICUB_LANE: the lane-group kernel below 16384 envs, the pipeline from there on"""
    from pybullet_robot_envs.model.table import icub_table
    monkeypatch.delenv("PBRE_ICUB_LANE", raising=False)
    tbl, model, info = icub_table("l")
    ov = parity.icub_overrides(info, "l", 1, 0, 1)
    for n, lane in ((64, 0), (16384, 1)):
        eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, lib=hip_lib, robot=_capi.ROBOT_ICUB, **ov)
        assert eng.kernel_info()[2] == lane, (n, eng.kernel_info())
        del eng


def test_icub_pipeline_at_size(hip_lib, monkeypatch):
    """32768 iCub-push envs through the lane-per-env pipeline (its default size range), Cartesian control, the hand driven down and towards
    the object so that table and object contacts occur.  Without pose randomisation and with one action for all envs every env is a
    replica: all 32768 rows must stay bit-identical through simple and complex steps (kw_quad / kw_quad_rc place an env anywhere in a
    wave or a list), finite, with unit quaternions.  (Agreement with the oracle is the other tests' business.)"""
    monkeypatch.delenv("PBRE_ICUB_LANE", raising=False)           # the engine's own choice at this size
    n = 32768
    eng, ora, info = parity.make_icub_pair(_capi.Engine, hip_lib, n, task=1, control_arm="l", use_ik=1, control_orientation=0,
                                           obj_std=0.0, tg_std=0.0, max_steps=60, flags=2)
    assert eng.kernel_info()[2] == 1
    obs = eng.reset()
    o0 = 9 + (eng.obs_dim - 9 - 15)
    seen_complex = 0
    for k in range(90):
        d = obs[0, o0:o0 + 3] - obs[0, 0:3]; d[2] -= 0.01
        a1 = np.clip(d / (np.linalg.norm(d) + 1e-6), -1, 1).astype(np.float32)
        obs, rw, dn = eng.step(np.broadcast_to(a1, (n, 3)).copy())
        assert np.isfinite(obs).all() and np.isfinite(rw).all(), k
        assert np.array_equal(obs, np.broadcast_to(obs[0], obs.shape)) and np.array_equal(rw, np.broadcast_to(rw[0], rw.shape)), "replicas diverged at step %d" % k
        seen_complex = max(seen_complex, eng.kernel_info()[5])
    st = eng.get_state(); nd = eng.ndof
    assert np.abs(np.linalg.norm(st[:, nd + 3:nd + 7], axis=1) - 1).max() < 1e-5
    assert seen_complex == n, "the hand never reached the object: the coupled solve was not exercised (%d)" % seen_complex
    print("iCub pipeline at size: all %d replicas identical over 90 steps, robot-object contact in every env at some step" % n)


def test_icub_hand_on_table(hip_lib):
    """robot-table contact rows of the lane-per-env pipeline (kw_quad) against the oracle, step by step"""
    print(parity.check_icub_table_contact(_capi.Engine, hip_lib, n=40, steps=60))


def test_icub_full_model_one_env_per_wave(hip_lib):
    parity.check_icub_full_model(_capi.Engine, hip_lib, n=5, steps=4)


def test_icub_wave_neighbour_independence(hip_lib):
    parity.check_wave_neighbour_independence(_capi.Engine, hip_lib)


def test_icub_reset_snapshot(hip_lib):
    import test_emu_icub
    test_emu_icub.test_icub_reset_snapshot(hip_lib)


# ---------------------------------------------------------------------------------------------- iCubEnv used alone
@pytest.mark.parametrize("arm,use_ik,ori", [("l", 0, 1), ("l", 1, 1), ("r", 1, 0)])
def test_icub_env_robot_level_commands(hip_lib, arm, use_ik, ori):
    """iCubEnv.apply_action(action, max_vel) + stepSimulation loops on the stand-alone class (csrc/pbre_icub_arm.hip: ShapeIA, half-wave
    lane groups with persistent motor records) through the C-ABI against the fp64 oracle"""
    parity.check_icub_arm(_capi.Engine, hip_lib, arm, use_ik, ori, n=5, steps=4)


def test_icub_env_robot_level_batch(hip_lib):
    """4096 replicas driven by a script: hand to a pose above the table with a velocity bound, then back; every env reaches the poses,
    replicas stay identical"""
    from pybullet_robot_envs import _client
    from pybullet_robot_envs.envs.icub_envs.icub_env import iCubEnv
    n = 4096
    cid = _client.connect(n, lib=hip_lib)
    robot = iCubEnv(cid, use_IK=1, control_arm='l', control_orientation=0)
    robot.apply_action([0.35, 0.2, 0.85], max_vel=2.0); robot.step_simulation(300)
    obs, lim = robot.get_observation()
    assert obs.shape == (n, 19) and np.abs(obs[:, :3] - [0.35, 0.2, 0.85]).max() < 5e-3, np.abs(obs[:, :3] - [0.35, 0.2, 0.85]).max()
    robot.apply_action([0.3, 0.26, 0.8]); robot.step_simulation(300)
    obs, lim = robot.get_observation()
    assert np.abs(obs[:, :3] - [0.3, 0.26, 0.8]).max() < 5e-3
    st = robot._client.engine.get_state()
    assert np.isfinite(st).all() and np.array_equal(st, np.broadcast_to(st[0], st.shape))
    assert robot.get_object_pose().shape == (n, 7) and robot.get_joint_positions().shape == (n, 20)
    _client.disconnect(cid)


@pytest.mark.parametrize("lane", ["1", "0"])
def test_icub_crafted_contact_states(hip_lib, monkeypatch, lane):
    """48 crafted contact states (hand on the object, on the table, both, a joint beyond its limit) through the lane-per-env pipeline
    (kw_quad / kw_quad_rc; lane = 1) and the lane-group kernel (lane = 0): one step each against the oracle, per-quantity bounds."""
    monkeypatch.setenv("PBRE_ICUB_LANE", lane)
    rep = parity.check_icub_contact_states(_capi.Engine, hip_lib, n_each=12)
    print("iCub crafted contact states (PBRE_ICUB_LANE=%s):" % lane, rep)
    assert rep["states"] == 48


@pytest.mark.parametrize("obj_name,lane", [("YcbTennisBall", "1"), ("YcbMasterChefCan", "1"), ("duck_vhacd", "1"), ("YcbMasterChefCan", "0")])
def test_icub_crafted_contact_states_round_objects(hip_lib, monkeypatch, obj_name, lane):
    """the crafted contact states with a round object (sphere / cylinder primitive) through the pipeline (lane = 1: hand on the ball /
    can / duck is kw_quad_rc's coupled solve with Fast::sphere_obj rows, the object's own rows are ObjStep's shape candidates) and
    through the lane-group kernel (lane = 0: what batches below 16384 envs run)"""
    monkeypatch.setenv("PBRE_ICUB_LANE", lane)
    rep = parity.check_icub_contact_states(_capi.Engine, hip_lib, n_each=12, obj_name=obj_name)
    print("iCub crafted contact states, %s (PBRE_ICUB_LANE=%s):" % (obj_name, lane), rep)
    assert rep["states"] == 48 and (rep["complex_envs_stepped"] > 0) == (lane == "1")


def test_icub_reach_default_object_runs_the_pipeline(hip_lib):
    """iCubReach-v0's default object (the duck: a round primitive) must not push the batch back to the lane-group kernel"""
    from pybullet_robot_envs.envs import iCubReachGymEnv
    env = iCubReachGymEnv(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, num_envs=16384, _lib=hip_lib)
    env.reset()
    assert env._engine.get_physics().obj_shape != 0 and env._engine.kernel_info()[2] == 1
    env.close()


def test_icub_push_closed_loop_against_oracle(hip_lib):
    rep = parity.check_icub_push_closed_loop(_capi.Engine, hip_lib, n=8)
    print("iCub closed-loop push:", rep)
    assert rep["touched_envs"] == 8


@pytest.mark.parametrize("use_ik", [0, 1])
def test_two_identical_pipelines_stay_bit_identical(hip_lib, monkeypatch, use_ik):
    """two identical iCub push engines side by side (pipeline, crafted contact states among 4096 envs, auto-reset): the envs with
    robot-object contact are stepped from a list whose order depends on timing (atomic appends); their results must not"""
    monkeypatch.setenv("PBRE_ICUB_LANE", "1")
    n = 4096
    eng0, ora, info = parity.make_icub_pair(_capi.Engine, hip_lib, 1, task=1, use_ik=0, obj_std=0.0, tg_std=0.2)
    base, _ = ora.batch_reset(1)
    S, kinds = parity.icub_contact_states(ora, info, base[0], np.random.default_rng(21), 12, 12, 12, 12, "l")
    kw = dict(task=1, use_ik=use_ik, obj_std=0.05, tg_std=0.2, max_steps=60, flags=_capi.F_AUTO_RESET)
    a, _, _ = parity.make_icub_pair(_capi.Engine, hip_lib, n, **kw)
    b, _, _ = parity.make_icub_pair(_capi.Engine, hip_lib, n, **kw)
    a.reset(); b.reset()
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S.astype(np.float32)
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(3)
    seen = 0
    for k in range(80):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        for x, y in zip(a.step(act), b.step(act)):
            assert np.array_equal(x, y)
        seen = max(seen, a.kernel_info()[5])
    assert np.array_equal(a.get_state(), b.get_state()) and seen > 0



@pytest.mark.parametrize("lane,use_ik", [("1", 0), ("1", 1), ("0", 0)])
def test_icub_nan_inf_guard(hip_lib, monkeypatch, lane, use_ik):
    """NaN / Inf guard on the iCub: the pipeline (kw_dyn records what the incoming state held, kw_fin / Lane::finish count, flag and restart)
    and the lane-group kernel (Core::step / Core::observe)"""
    monkeypatch.setenv("PBRE_ICUB_LANE", lane)
    parity.check_icub_nan_guard(_capi.Engine, hip_lib, use_ik=use_ik)


@pytest.mark.parametrize("task", [0, 1])
def test_icub_solver_residual_threshold(hip_lib, task):
    """pbre_physics.solver_residual_threshold on the iCub (kw_step<.., RT>: the lane-group kernel steps the batch, the lane-per-env pipeline
    splits an env's rows over kernels and is switched off): joint-control steps against the oracle with the same threshold."""
    eng, ora, info = parity.make_icub_pair(_capi.Engine, hip_lib, 8, task, "l", 0, 0, obj_std=0.05, tg_std=0.2)
    eng.reset(); st, _ = ora.batch_reset(8)
    parity.check_group_residual_threshold(eng, ora, st, np.random.default_rng(5), parity.TOL_ICUB, steps=3)
    eng.set_physics(solver_residual_threshold=1e-7)
    assert eng.kernel_info()[2] == 0          # with the threshold on, the lane-group kernel steps the batch (pbre_lane.hip: lane_ok)


@pytest.mark.parametrize("use_ik", [0, 1])
def test_icub_floating_base_option(hip_lib, use_ik):
    """The soft-pinned floating base (model/table.py: float_base; 26 DoF, kw_step<Shape64>) on the device against the oracle, and its
    effect on the hand's observation against the rigidly pinned default model."""
    rep = parity.check_icub_floating_base(_capi.Engine, hip_lib, n=4, steps=3, use_ik=use_ik)
    assert rep["ee_pos_shift_vs_pinned_base_40_steps_m"] < 3e-2      # (12 mm with the constraint's 500 N bound on its rows, round 6; 2 mm unbounded)


def test_icub_base_constraint_force_bound(hip_lib):
    """the base constraint's maxForce as a bound on its rows, set below the robot's weight: the robot sinks at the free-fall deficit, device
    against oracle"""
    rep = parity.check_icub_base_force_bound(_capi.Engine, hip_lib, n=4)
    print(rep)


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


@pytest.mark.gpu
def test_icub_hull_object_from_a_mesh_file(hip_lib, tmp_path, monkeypatch):
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
    rep = parity.check_icub_contact_states(_capi.Engine, hip_lib, n_each=12, obj_name="duck_vhacd")
    assert rep["states"] == 4 * 12 and rep["complex_envs_stepped"] >= 0
