"""Known-answer tests that pin the oracle without PyBullet.  CPU only."""
import numpy as np
import pytest

import orc

HOME = np.array([0, -0.54, 0, -2.6, -0.30, 2.0, 1.0, 0.02, 0.02])


@pytest.fixture(scope="module")
def ora(panda):
    return orc.Oracle(panda["table"], task=1)


def test_k1_forward_kinematics(ora):
    # zero and home configuration of panda_grasptarget / hand / link7 / fingers
    R, p = ora.fk(np.zeros(9))
    assert np.allclose(p[11], [0.088, 0.0, 1.481], atol=1e-6)
    R, p = ora.fk(HOME)
    assert np.allclose(p[11], [0.36586, -0.03674, 0.986768], atol=1e-6)
    assert np.allclose(R[11], [[0.996343, -0.075029, -0.040879], [-0.083288, -0.959612, -0.268716],
                               [-0.019067, 0.271138, -0.962352]], atol=1e-6)
    assert np.allclose(p[8], [0.368721, -0.01793, 1.054133], atol=1e-6)
    assert np.allclose(p[6], [0.373095, 0.010822, 1.157104], atol=1e-6)
    assert np.allclose(p[9], [0.364833, -0.052816, 1.003354], atol=1e-6)
    assert np.allclose(p[10], [0.367834, -0.014431, 0.992509], atol=1e-6)


def _jacobians(ora, q):
    m = ora.model
    R, p = ora.fk(q)
    Jv = np.zeros((ora.nl, 3, ora.ndof)); Jw = np.zeros((ora.nl, 3, ora.ndof)); com = np.zeros((ora.nl, 3))
    for i in range(ora.nl):
        com[i] = p[i] + R[i] @ np.array(m.com[i])
        k = i
        while k >= 0:
            if m.jtype[k] != 0:
                aw = R[k] @ np.array(m.axis[k]); d = m.dof[k]
                if m.jtype[k] == 1:
                    Jv[i, :, d] = np.cross(aw, com[i] - p[k]); Jw[i, :, d] = aw
                else:
                    Jv[i, :, d] = aw
            k = m.parent[k]
    return R, Jv, Jw, com


def test_aba_matches_composite_mass_matrix(ora):
    # independent derivation: M = sum m Jv^T Jv + Jw^T I Jw ; gravity torque g = sum m g Jv_z
    rng = np.random.default_rng(0)
    m = ora.model
    for _ in range(4):
        q = rng.uniform(-1, 1, 9); q[7:] = rng.uniform(0, 0.04, 2)
        R, Jv, Jw, com = _jacobians(ora, q)
        M = np.zeros((9, 9)); g = np.zeros(9)
        for i in range(ora.nl):
            Iw = R[i] @ np.array(m.inertia[i]).reshape(3, 3) @ R[i].T
            M += m.mass[i] * Jv[i].T @ Jv[i] + Jw[i].T @ Iw @ Jw[i]
            g += m.mass[i] * 9.8 * Jv[i][2]
        assert np.abs(np.linalg.inv(M) - ora.minv(q)).max() < 1e-10
        assert np.abs(ora.forward_dynamics(q, np.zeros(9)) - np.linalg.solve(M, -g)).max() < 1e-9


def test_aba_conserves_energy_without_damping(panda):
    o = orc.Oracle(panda["table"])
    o.params.lin_damping = 0; o.params.ang_damping = 0
    m = o.model
    rng = np.random.default_rng(1)

    def energy(q, qd):
        R, Jv, Jw, com = _jacobians(o, q)
        E = 0.0
        for i in range(o.nl):
            Iw = R[i] @ np.array(m.inertia[i]).reshape(3, 3) @ R[i].T
            v = Jv[i] @ qd; w = Jw[i] @ qd
            E += 0.5 * m.mass[i] * v @ v + 0.5 * w @ Iw @ w + m.mass[i] * 9.8 * com[i][2]
        return E

    q = rng.uniform(-1, 1, 9); q[7:] = 0.02
    qd = rng.uniform(-2, 2, 9); qd[7:] *= 0.05
    y = np.concatenate([q, qd]); E0 = energy(q, qd); h = 1e-3
    f = lambda y: np.concatenate([y[9:], o.forward_dynamics(y[:9], y[9:])])
    for _ in range(100):
        k1 = f(y); k2 = f(y + h / 2 * k1); k3 = f(y + h / 2 * k2); k4 = f(y + h * k3)
        y = y + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    assert np.abs(y[:9] - q).max() > 0.1            # the arm really moved
    assert abs(energy(y[:9], y[9:]) - E0) < 1e-6 * abs(E0)


def test_philox_known_answers(ora):
    # Random123 kat_vectors for philox4x32-10
    assert ora.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ora.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ora.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_k3_k6_reset_rest_pose(ora):
    st, obs = ora.batch_reset(1)
    s = st[0]
    assert np.allclose(s[:9], HOME, atol=1e-6)                     # motors hold the home pose
    assert np.allclose(s[9:11], [0.45, 0.0], atol=1e-6)            # K6 object start x,y
    assert abs(s[11] - 0.650) < 1e-4                               # K3 cube rests at h + 0.025
    assert np.allclose(s[12:16], [0, 0, 0.382683, 0.923880], atol=1e-5)
    assert np.abs(s[25:31]).max() < 1e-3
    assert np.allclose(s[32:35], [0.5, 0.05, s[11]], atol=1e-6)    # target = obj + (0.05, 0.05, 0)
    assert obs.shape == (1, 33)


def test_k2_motor_law_and_k4_first_step(ora):
    st, _ = ora.batch_reset(1)
    a = np.array([[1, -1, 0.5, -0.5, 1, -1, 0.3]])
    st2, out = ora.batch_step(st, a)
    assert np.allclose(st2[0, :7] - st[0, :7], 0.025 * a[0], atol=1e-6)     # K2: dq = kp * 0.05 * a
    assert np.allclose(st2[0, 16:23], 0.025 * a[0] * 240, atol=1e-3)
    assert np.allclose(st2[0, 7:9], 0.02, atol=1e-6)
    # K4 (quirk E-1): default target is inside the success radius -> done, reward 1000 + (100 - 80 d)
    d = np.linalg.norm(st2[0, 9:12] - st2[0, 32:35])
    assert out[0, -1] == 1.0 and abs(out[0, -2] - (1100 - 80 * d)) < 1e-9
    assert abs(out[0, -2] - 1094.343) < 1e-2


def test_termination_counter_semantics(ora):
    ora.task.tg_pose_rnd_std = 0.0
    st, _ = ora.batch_reset(1)
    st[0, 32:35] = [0.6, 0.3, st[0, 11]]          # far target: no success
    ora.task.max_steps = 3
    dones = []
    for _ in range(6):
        st, out = ora.batch_step(st, np.zeros((1, 7)))
        dones.append(out[0, -1]); 
    # counter > max_steps is evaluated before the increment inside apply_action and after it in step()
    assert dones == [0, 0, 0, 1, 1, 1]
    assert st[0, 35] == 4
    ora.task.max_steps = 1000


def test_object_stays_on_table_and_is_pushable(ora):
    # sweep the hand through the cube slowly: the cube is pushed along +x, stays on the table, nothing blows up
    import scenarios
    ora.task.tg_pose_rnd_std = 0.0
    st, _ = ora.batch_reset(1)
    st[0, 32:35] = [0.6, 0.3, st[0, 11]]
    q_pre, q_end, n1, n2 = scenarios.push_actions(ora, st[0])
    x0 = st[0, 9:12].copy()
    for t in range(n1 + 700):
        a = scenarios.track(st[0], q_pre if t < n1 else q_end, 1.0 if t < n1 else 0.1)[None]
        st, out = ora.batch_step(st, a)
        assert np.isfinite(st).all()
        assert 0.64 < st[0, 11] < 0.67                    # stays on the table top (rest height 0.650), may tilt a little
    assert st[0, 9] - x0[0] > 0.03                       # pushed forward (the joint-space sweep also turns it)
    assert abs(st[0, 10] - x0[1]) < 0.08


def test_sliding_cube_stops_where_coulomb_friction_says(ora):
    """analytic contact KAT: stopping distance of a sliding cube (parity.check_sliding_cube_kat) -- pins the oracle's friction rows,
    normal force and damping against first principles, not against PyBullet"""
    import parity
    ora.task.tg_pose_rnd_std = 0.0; ora.task.obj_pose_rnd_std = 0.0
    st, _ = ora.batch_reset(1)
    P = ora.params
    zero = np.zeros((1, 7))
    rep = parity.check_sliding_cube_kat(lambda s: ora.batch_step(s, zero)[0], st,
                                        {"mu": P.obj_mu * P.table_mu, "g": -P.gravity_z, "kl": P.lin_damping, "dt": P.dt})
    assert rep["axis_v0.6"]["distance_m"] > 0.03
    ff = parity.check_free_fall_kat(lambda s: ora.batch_step(s, zero)[0], st, {"g": -P.gravity_z, "kl": P.lin_damping, "dt": P.dt})
    assert ff["fall_m"] > 0.05


def test_sliding_ball_and_can_end_up_rolling_at_the_analytic_speed(panda):
    """analytic KAT for the round primitives (parity.check_rolling_onset_kat): 5/7 v0 for the ball, 2/3 v0 for the lying can"""
    import parity
    from pybullet_robot_envs.model.objects import object_physics

    def make(name):
        ph = object_physics(name)
        o = orc.Oracle(panda["table"], task=1)
        orc.set_object(o, ph)
        o.task.tg_pose_rnd_std = 0.0; o.task.obj_pose_rnd_std = 0.0
        st, _ = o.batch_reset(1)
        zero = np.zeros((1, 7))
        return st, ph["obj_h"][0], (lambda s: o.batch_step(s, zero)[0])
    rep = parity.check_rolling_onset_kat(make)
    assert rep["YcbTennisBall"]["steps_to_rolling"] > 5



def test_oracle_under_asan_ubsan():
    """SURVEY section 5 (race / memory-error detection: the reference has none): the oracle's own tests once more against its
    AddressSanitizer + UndefinedBehaviorSanitizer build (`make -C oracle asan`), in a subprocess with the sanitizer runtime preloaded.
    Any out-of-bounds access, use of an uninitialised stack slot through a wild index, signed overflow or misaligned access in the
    restatement aborts that run."""
    import os, subprocess, sys
    if os.environ.get("ORC_SANITIZED") == "1":
        return                                   # (we ARE the sanitized re-run)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ub = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not (os.path.isabs(rt) and os.path.exists(rt)):
        import pytest
        pytest.skip("no libasan in this image")
    env = dict(os.environ, ORC_SANITIZED="1", LD_PRELOAD=rt + (":" + ub if os.path.exists(ub) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail


def test_solver_residual_threshold_option():
    """Bullet's exit test of the sweep loop (orc_params.solver_residual_threshold; PyBullet documents solverResidualThreshold with default
    1e-7 [EXT-UNVERIFIED]): off by default -- all solver_iters sweeps, which is what the engine implements --, and when it is on the loop
    ends exactly where orc_step_info.sweeps_to_1e7 says, with a result close to the full solve's (tools/residual_exit_probe.py)."""
    import orc
    ora, _ = orc.panda_oracle()
    ora.task.obj_pose_rnd_std = 0.05; ora.task.tg_pose_rnd_std = 0.2
    n = 6
    st, _ = ora.batch_reset(n)
    act = np.random.default_rng(3).uniform(-1, 1, (n, 7))
    assert ora.params.solver_residual_threshold == 0.0
    full, out_full, used, to7 = ora.batch_step_sweeps(st, act)
    assert (used == ora.params.solver_iters).all() and (to7 >= 1).all()
    ora.params.solver_residual_threshold = 1e-7
    early, out_early, used_e, to7_e = ora.batch_step_sweeps(st, act)
    ora.params.solver_residual_threshold = 0.0
    assert (used_e == np.minimum(to7, ora.params.solver_iters)).all() and (to7_e == to7).all()
    assert (used_e < ora.params.solver_iters).any(), "the test never fires on free-space steps: nothing tested"
    nd = ora.ndof
    assert np.abs(early[:, :nd] - full[:, :nd]).max() < 1e-5 and np.abs(early[:, 16:16 + nd] - full[:, 16:16 + nd]).max() < 2e-3
    # bit-identical to the default when nothing ends the loop early
    again, _, _, _ = ora.batch_step_sweeps(st, act)
    assert np.array_equal(again, full)
