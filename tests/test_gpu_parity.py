"""Parity tests proper: the HIP engine (libpbre.so, through the C-ABI) vs the CPU oracle on a real
MI355X.  Same checks as test_emu_parity.py plus full-size property tests."""
import numpy as np
import pytest

import parity
from pybullet_robot_envs import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["pair", "lane"])
def simple_env_mapping(request, monkeypatch):
    """Every test of this module runs under both mappings of the simple class: the pair kernel (k_fast_pair: robot wave + object wave per
    64 envs -- what launch_step picks by itself for batches of up to 65536 envs, i.e. for every per-GPU shard of BASELINE's split) and
    the one-lane-per-env kernel k_fast (what the full 131072-env batch gets).  PBRE_PAIR is read at pbre_create."""
    monkeypatch.setenv("PBRE_PAIR", "1" if request.param == "pair" else "0")
    return request.param


@pytest.mark.parametrize("task", [0, 1])
def test_reset_and_steps(panda, hip_lib, task):
    n = 50                                   # not a multiple of 16: exercises the padding rows
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], n, task=task)
    st = parity.check_reset(eng, ora, n)
    parity.check_single_steps(eng, ora, st, np.random.default_rng(0), steps=6)


@pytest.mark.parametrize("flags", [0, _capi.F_COMPLEX_ROWS, _capi.F_COMPLEX_LANES])
def test_contact_rich_states(panda, hip_lib, flags):
    """flags: which kernel steps the envs with robot contacts (0: the engine's choice by their number, F_COMPLEX_ROWS: k_row_list,
    F_COMPLEX_LANES: k_fast_rc)"""
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    rng = np.random.default_rng(1)
    S = parity.contact_states(ora, panda, base[0], rng, 24, 24)
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], len(S), flags=flags)
    parity.check_single_steps(eng, ora, S, rng, steps=1, tol=parity.TOL_CONTACT, skip_ambiguous=True, max_skip=0.1)


@pytest.mark.parametrize("flags", [_capi.F_COMPLEX_ROWS, _capi.F_COMPLEX_LANES])
def test_joint_limit_rows(panda, hip_lib, flags):
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 4, flags=flags)
    st, _ = ora.batch_reset(4)
    st[0, 3] = 0.02
    st[1, 5] = -0.12
    st[2, 7] = 0.045
    st[3, 1] = -1.9
    parity.check_single_steps(eng, ora, st, np.random.default_rng(2), steps=2)


def test_free_running_rollout(panda, hip_lib):
    n = 32
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], n)
    st = parity.check_reset(eng, ora, n)
    rng = np.random.default_rng(3)
    for _ in range(60):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        ob, rw, dn = eng.step(a)
        st, out = ora.batch_step(st, a)
    parity.assert_within(parity.panda_quantities(eng.get_state(), st, ob, out, rw), parity.TOL_ROLLOUT60, "(60 free-running steps after reset)")


def test_full_episode_rollout(panda, hip_lib):
    """A whole 1000-step episode, free running, against the oracle: drift bounds stated in parity.check_panda_full_episode."""
    w = parity.check_panda_full_episode(_capi.Engine, hip_lib, panda["table"], n=24, steps=1000)
    print("full-episode drift:", w)


def test_device_glue_only(panda, hip_lib):
    """The observation glue alone on the GPU from reference-captured states: <= 1e-6."""
    print("glue-only:", parity.check_device_glue(_capi.Engine, hip_lib, panda["table"]))


def test_matches_host_lane_emulation(panda, hip_lib, emu_lib):
    """The CPU lane emulation compiles the same source (csrc/pbre_fast.hpp, pbre_core.hpp) without FMA contraction and with libm's
    division / square root, so it is not bit-identical to the device; one step from the same state agrees per quantity within a
    quarter of the single-step bounds against the oracle (parity.TOL), the 201-step reset within 1e-5 (positions, angles)."""
    n = 20
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    g = _capi.Engine(panda["table"], lib=hip_lib, **kw)
    e = _capi.Engine(panda["table"], lib=emu_lib, **kw)
    og, oe = g.reset(), e.reset()
    assert np.abs(g.get_state()[:, :16] - e.get_state()[:, :16]).max() < 1e-5 and np.abs(og - oe).max() < 1e-4
    e.set_state(g.get_state())
    a = np.random.default_rng(6).uniform(-1, 1, (n, 7)).astype(np.float32)
    (obg, rwg, dng), (obe, rwe, dne) = g.step(a), e.step(a)
    out_e = np.concatenate([obe, rwe[:, None], dne[:, None]], 1).astype(np.float64)
    q = parity.panda_quantities(g.get_state(), e.get_state().astype(np.float64), obg, out_e, rwg)
    parity.assert_within(q, dict((k, 0.25 * v) for k, v in parity.TOL.items()), "(GPU against the CPU lane emulation)")
    assert np.array_equal(dng, dne)


def test_sharding_invariance(panda, hip_lib):
    kw = dict(task=1, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib)
    full = _capi.Engine(panda["table"], num_envs=96, **kw)
    a = _capi.Engine(panda["table"], num_envs=48, env_id_base=0, **kw)
    b = _capi.Engine(panda["table"], num_envs=48, env_id_base=48, **kw)
    of, oa, ob = full.reset(), a.reset(), b.reset()
    assert np.array_equal(of, np.concatenate([oa, ob]))
    act = np.random.default_rng(4).uniform(-1, 1, (96, 7)).astype(np.float32)
    for _ in range(3):
        rf = full.step(act); ra = a.step(act[:48]); rb = b.step(act[48:])
        for x, y, z in zip(rf, ra, rb):
            assert np.array_equal(x, np.concatenate([y, z]))


def test_masked_reset(panda, hip_lib):
    n = 40
    eng = _capi.Engine(panda["table"], task=1, num_envs=n, lib=hip_lib, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    first = eng.reset()
    rng = np.random.default_rng(5)
    for _ in range(3):
        eng.step(rng.uniform(-1, 1, (n, 7)).astype(np.float32))
    before = eng.get_state()
    mask = np.zeros(n, np.uint8); mask[[1, 7, 33]] = 1
    eng.reset(mask=mask)
    after = eng.get_state()
    keep = mask == 0
    assert np.array_equal(before[keep], after[keep])
    assert (after[mask == 1, 35] == 0).all() and (after[mask == 1, 37] == 1).all()
    # a full reset of a fresh engine reproduces episode 0 exactly (deterministic streams)
    eng2 = _capi.Engine(panda["table"], task=1, num_envs=n, lib=hip_lib, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    assert np.array_equal(first, eng2.reset())


@pytest.mark.parametrize("n", [32768, 131072, 1048576 + 37])
def test_full_size_properties(panda, hip_lib, n):
    """BASELINE config 3 / config 4 sizes (32768, 131072 envs) and a ragged 1 M batch: size-independent properties."""
    eng = _capi.Engine(panda["table"], task=1, num_envs=n, lib=hip_lib, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    obs = eng.reset()
    st0 = eng.get_state()
    assert np.isfinite(st0).all() and np.isfinite(obs).all()
    assert np.abs(st0[:, 11] - 0.650).max() < 1e-3                       # every cube rests on the table (K3)
    assert np.abs(st0[:, 9] - 0.45).max() <= 0.05 + 1e-6                 # object x,y noise U(+-0.05)
    assert np.abs(st0[:, 10]).max() <= 0.05 + 1e-6
    assert 0.37 - 1e-6 <= st0[:, 32].min() and st0[:, 32].max() <= 0.58 + 1e-6   # target clip range (K6)
    assert st0[:, 32].std() > 0.05
    rng = np.random.default_rng(7)
    a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    ob, rw, dn = eng.step(a)
    st1 = eng.get_state()
    assert np.isfinite(st1).all() and np.isfinite(ob).all() and np.isfinite(rw).all()
    # K2 motor law in free space: dq = 0.5 * (clip(q + 0.05 a) - q)
    assert np.abs((st1[:, :7] - st0[:, :7]) - 0.025 * a).max() < 2e-5
    assert np.abs(st1[:, 7:9] - 0.02).max() < 1e-5
    # identical inputs -> identical outputs (replicated envs), no cross-env leakage
    eng.set_state(np.repeat(st0[:1], n, 0))
    ob2, rw2, dn2 = eng.step(np.repeat(a[:1], n, 0))
    assert (ob2 == ob2[0]).all() and (rw2 == rw2[0]).all()
    for _ in range(20):
        ob, rw, dn = eng.step(rng.uniform(-1, 1, (n, 7)).astype(np.float32))
    st = eng.get_state()
    assert np.isfinite(st).all()
    assert (st[:, 11] > 0.62).all()                                       # nothing fell through the table
    assert np.abs(np.linalg.norm(st[:, 12:16], axis=1) - 1).max() < 1e-5  # unit quaternions


def test_reference_golden_outputs(hip_lib):
    """Outputs of the reference's own classes (tests/golden, captured over the oracle's physics) reproduced by the
    HIP engine through the drop-in classes, step by step from the reference's states."""
    import os
    import test_golden_glue as tgg
    from pybullet_robot_envs.envs import pandaPushGymEnv, pandaReachGymEnv, pandaPushGymGoalEnv
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "panda_glue.npz"))
    for cls, tag, kw in [(pandaPushGymEnv, "pushA", {}), (pandaPushGymEnv, "pushB", {"max_steps": 6}),
                         (pandaReachGymEnv, "reachC", {"max_steps": 5}),
                         (pandaPushGymGoalEnv, "goalD", {"max_steps": 4, "tg_pose_rnd_std": 0.0})]:
        env = cls(**kw)
        goal = tag.startswith("goal")
        o = env.reset()
        o = o["observation"] if goal else o
        if tag != "goalD":
            assert np.abs(o - G[tag + "_reset_obs"]).max() < 2e-3
        for k in range(len(G[tag + "_actions"])):
            s = np.zeros((1, 48), np.float32)
            s[0] = G[tag + "_pre_state"][k]
            env._engine.set_state(s)
            ob, r, d, info = env.step(G[tag + "_actions"][k])
            ob = ob["observation"] if goal else ob
            tgg.check_scaled_obs(ob, G[tag + "_obs"][k], tag, k)
            assert abs(float(r) - G[tag + "_reward"][k]) < 2e-5 * max(1, abs(G[tag + "_reward"][k]))
            assert float(d) == G[tag + "_done"][k]
            assert int(env._env_step_counter) == G[tag + "_counter"][k]
        env.close()
    print("golden replay worst (scaled obs):", tgg.WORST)


@pytest.mark.parametrize("flags", [0, _capi.F_COMPLEX_LANES, _capi.F_FORCE_GENERAL])
def test_auto_reset(panda, hip_lib, flags):
    parity.check_auto_reset(_capi.Engine, hip_lib, panda["table"], n=200, max_steps=4, flags=flags)


def test_lane_per_env_kernels_match_row_kernel(panda, hip_lib):
    """k_fast / k_fast_rc (lane-per-env) against the general 16-lane row kernel (PBRE_F_FORCE_GENERAL) on the same states,
    including contact-rich and limit-violating ones."""
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    rng = np.random.default_rng(12)
    S = parity.contact_states(ora, panda, base[0], rng, 40, 40).astype(np.float32)
    S[3, 3] = 0.02; S[5, 5] = -0.12          # joint-limit rows
    n = len(S)
    a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    kw = dict(task=1, num_envs=n, lib=hip_lib)
    g = _capi.Engine(panda["table"], flags=_capi.F_FORCE_GENERAL, **kw)
    g.set_state(S)
    rg = g.step(a)
    amb = parity.ambiguous_envs(ora, S.astype(np.float64), a)
    ok = ~amb
    for fl in (_capi.F_COMPLEX_LANES, _capi.F_COMPLEX_ROWS):   # both kernels for the complex envs
        f = _capi.Engine(panda["table"], flags=fl, **kw)
        f.set_state(S)
        info = f.kernel_info()
        assert info[2] == 1 and info[5] >= 40                  # fast path enabled, most of these states are "complex"
        rf = f.step(a)
        assert parity.rel(f.get_state()[ok], g.get_state()[ok].astype(np.float64)).max() < 1e-3
        assert parity.rel(rf[0][ok], rg[0][ok].astype(np.float64)).max() < 5e-3


def test_env_classes_and_tensor_api(hip_lib):
    import torch
    from pybullet_robot_envs.envs import pandaPushGymEnv
    n = 256
    env = pandaPushGymEnv(num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    ref = pandaPushGymEnv(num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    o0 = env.reset(); ref.reset()
    assert o0.shape == (n, 33) and o0.dtype == np.float64
    a = np.random.default_rng(1).uniform(-1, 1, (n, 7)).astype(np.float32)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        obs_t, rew_t, done_t = env.step_tensor(torch.as_tensor(a, device="cuda"))
        torch.cuda.synchronize()
    obs_h, rew_h, done_h, _ = ref.step(a)
    assert np.abs(obs_t.cpu().numpy() - obs_h).max() < 1e-4          # float32 scaling on device vs float64 on host
    assert np.allclose(rew_t.cpu().numpy(), rew_h, atol=1e-5) and np.array_equal(done_t.cpu().numpy(), done_h)
    env.change_physics_params(0.2, 0.7, 0.05, 0.02)                # object: mass, friction, damping; robot damping separately
    assert np.allclose(env._engine.get_state_cols(44, 4)[:, [0, 1, 3]], [0.2, 0.7, 1.05]) and np.allclose(env._engine.get_state_cols(31, 1), 1.02)
    env.close(); ref.close()


@pytest.mark.parametrize("flags", [0, _capi.F_FORCE_GENERAL])
def test_other_objects(panda, hip_lib, flags):
    """obj_name other than the cube (YCB / pybullet_data box stand-ins): k_fast + ObjStep and k_row_list, and the general row kernel"""
    print(parity.check_other_objects(_capi.Engine, hip_lib, panda["table"], n=40, flags=flags))


@pytest.mark.parametrize("flags", [0, _capi.F_FORCE_GENERAL])
def test_reset_snapshot(panda, hip_lib, flags):
    parity.check_reset_snapshot(_capi.Engine, hip_lib, panda["table"], n=70, flags=flags)


def test_devices_kwarg(hip_lib):
    """Gym classes with devices=[...]: one pbre_ctx per listed device driven from one process (two shards on this box's one GPU)."""
    import test_vec_env
    test_vec_env.test_devices_kwarg_shards_the_batch_in_one_process(hip_lib)


@pytest.mark.parametrize("flags", [0, _capi.F_COMPLEX_LANES, _capi.F_FORCE_GENERAL])
def test_per_env_domain_randomisation(panda, hip_lib, flags):
    """per-env object mass / friction / damping against the oracle (k_fast, k_fast_rc / k_row_list, general row kernel)"""
    print(parity.check_per_env_physics(_capi.Engine, hip_lib, panda["table"], n=70, flags=flags)["worst"])


def test_step_tensor_is_ordered_with_the_producing_stream(hip_lib):
    """on torch's default (null) stream the actions may still be in flight when step_tensor is called.  The
    step must be enqueued in stream order behind the kernels that produce the actions and ahead of kernels that overwrite them;
    host-synchronous entry points called right after must see the finished step."""
    import torch
    from pybullet_robot_envs.envs import pandaPushGymEnv
    n = 4096
    env = pandaPushGymEnv(num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    ref = pandaPushGymEnv(num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    env.reset(); ref.reset()
    a = np.random.default_rng(2).uniform(-1, 1, (n, 7)).astype(np.float32)
    a_dev = torch.as_tensor(a, device="cuda")
    torch.cuda.synchronize()
    for stream in (None, torch.cuda.Stream()):
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream()):
            x = torch.randn(2048, 2048, device="cuda")
            for _ in range(30):                        # tens of ms of queued work ahead of the actions
                x = torch.tanh(x @ x) * 0.01
            act = a_dev + 0.0 * x[0, 0]                # produced at the end of that chain, asynchronously
            obs_t, rew_t, done_t = env.step_tensor(act)
            act.fill_(7.0)                             # overwritten right after the step was enqueued
            st = env._engine.get_state()               # host-synchronous entry point right behind an external-stream step
        obs_h, rew_h, done_h, _ = ref.step(a)
        assert np.array_equal(st, ref._engine.get_state())
        assert np.abs(obs_t.cpu().numpy() - obs_h).max() < 1e-4
        assert np.allclose(rew_t.cpu().numpy(), rew_h, atol=1e-5) and np.array_equal(done_t.cpu().numpy(), done_h)
    # returned reward / done tensors are the caller's: a later step does not overwrite them
    keep = rew_t.clone()
    env.step_tensor(a_dev * 0.5)
    torch.cuda.synchronize()
    assert torch.equal(keep, rew_t)
    env.close(); ref.close()


def test_sharding_invariance_at_the_headline_size(panda, hip_lib, simple_env_mapping, monkeypatch):
    """BASELINE config 4 as a size-independent property: 131072 envs in ONE engine (k_fast, 2- and 3-wave builds, k_row_list) against the
    same envs in two 65536-env shards with env_id_base 0 / 65536 (k_fast_pair steps their simple envs), the bench's stationary protocol
    (de-synchronised episode clocks, i.i.d. actions, in-kernel auto-reset), 600 steps: output rows and final states bit for bit
    (tools/diag_sharding_at_scale.py runs the same for 1500 steps)."""
    import torch
    if simple_env_mapping == "lane":
        pytest.skip("one run: this test leaves the choice of the simple envs' kernel to the library")
    monkeypatch.setenv("PBRE_PAIR", "2")          # the default: the pair kernel while its waves fit two per SIMD
    n, h, steps = 131072, 65536, 600
    kw = dict(task=1, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET, seed=1234, lib=hip_lib)
    whole = _capi.Engine(panda["table"], num_envs=n, **kw)
    parts = [_capi.Engine(panda["table"], num_envs=h, env_id_base=k * h, **kw) for k in range(2)]
    clocks = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
    for e, sl in [(whole, slice(0, n))] + [(parts[k], slice(k * h, (k + 1) * h)) for k in range(2)]:
        e.reset()
        st = e.get_state(); st[:, e.x_off + 3] = clocks[sl]; e.set_state(st)
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(device=dev)
    ow = whole.obs_dim + 2
    out_w = torch.zeros((n, ow), device=dev); out_p = torch.zeros((n, ow), device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    with torch.cuda.stream(s):
        for k in range(steps):
            act = torch.rand((n, 7), device=dev, generator=gen) * 2 - 1
            whole.step_device(act.data_ptr(), out_w.data_ptr(), s.cuda_stream)
            for j in range(2):
                parts[j].step_device(act[j * h:(j + 1) * h].data_ptr(), out_p[j * h:(j + 1) * h].data_ptr(), s.cuda_stream)
            if k % 100 == 99:
                s.synchronize()
                assert torch.equal(out_w, out_p), "output rows differ at step %d" % k
    assert np.array_equal(whole.get_state(), np.concatenate([p.get_state() for p in parts]))
    iw, ip = whole.kernel_info(), parts[0].kernel_info()
    assert ip[10] >= steps and iw[10] == 0, "the shards are stepped by the pair kernel, the whole batch by k_fast"
    assert iw[7] > 2 * steps and iw[12] == 0 and ip[12] == 0       # complex envs throughout (8.4 k env-steps of them in the run of record); no NaN / Inf
    print("sharding at scale: complex env-steps %d, 3-wave steps %d, pair steps per shard %d" % (iw[7], iw[8], ip[10]))


def test_config2_reach_without_object(panda, hip_lib):
    """BASELINE config 2: Panda reach, 4096 envs, object frozen and contact-free (free-space dynamics only)."""
    n = 4096
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], n, task=0, obj_std=0.05, tg_std=0.0, flags=_capi.F_NO_OBJECT)
    import orc as _orc
    ora.params.flags = _orc.F_NO_OBJECT
    eng.reset()
    st0 = eng.get_state()
    rng = np.random.default_rng(8)
    a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    ob, rw, dn = eng.step(a)
    st1 = eng.get_state()
    so, out = ora.batch_step(st0.astype(np.float64), a)          # every env of the batch against the oracle, per quantity
    q = parity.panda_quantities(st1, so, ob, out, rw, task=0)
    parity.assert_within(q, parity.TOL, "(config 2: one step of all 4096 reach envs)")
    assert np.array_equal(dn, out[:, -1])
    assert np.array_equal(st1[:, 9:16], st0[:, 9:16])                           # the object did not move
    assert np.abs((st1[:, :7] - st0[:, :7]) - 0.025 * a).max() < 2e-5          # K2 motor law


@pytest.mark.parametrize("task,flags", [(1, 0), (0, 0), (1, _capi.F_FORCE_GENERAL)])
def test_ik_mode(panda, hip_lib, task, flags):
    """use_IK=1 (k_ik + target-mode step kernels) against the oracle, lane-per-env and row kernels."""
    eng = parity.check_ik_mode(_capi.Engine, hip_lib, panda["table"], task, flags=flags)
    info = eng.kernel_info()
    assert (info[2] == 1) == (flags == 0)


@pytest.mark.parametrize("n", [1, 5, 17, 63, 65, 130])
def test_ragged_batch_sizes(panda, hip_lib, n):
    """Batch sizes that are not multiples of the 16-env row blocks / 64-env waves, with the complex envs spread over them."""
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], n)
    st = parity.check_reset(eng, ora, n)
    st[::3, 3] = 0.02                      # every third env: joint 4 over its limit -> complex list, row kernel
    parity.check_single_steps(eng, ora, st, np.random.default_rng(n), steps=2)


@pytest.mark.parametrize("flags", [_capi.F_COMPLEX_ROWS, _capi.F_COMPLEX_LANES])
def test_scripted_push_properties(panda, hip_lib, flags):
    """A scripted closed-loop push (the hand sweeps through the cube, hundreds of steps in robot-object contact) on a batch
    with randomised object poses, with either complex-env kernel: the cube is pushed forward, stays on the table top,
    nothing blows up -- the properties tests/test_oracle.py asserts of the oracle for the same script."""
    import scenarios
    n = 48
    eng, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], n, obj_std=0.02, tg_std=0.0, flags=flags)
    eng.reset()
    st = eng.get_state().astype(np.float64)
    st[:, 32:35] = [0.6, 0.3, 0.65]                           # far target: no success latch during the script
    eng.set_state(st.astype(np.float32))
    plans = [scenarios.push_actions(ora, st[e]) for e in range(n)]
    n1 = max(p[2] for p in plans)
    x0 = st[:, 9:12].copy()
    seen_complex = 0
    for t in range(n1 + 500):
        a = np.stack([scenarios.track(st[e], plans[e][0] if t < n1 else plans[e][1], 1.0 if t < n1 else 0.1) for e in range(n)])
        eng.step(a.astype(np.float32))
        st = eng.get_state().astype(np.float64)
        if t % 50 == 0:
            seen_complex = max(seen_complex, eng.kernel_info()[5])
            assert np.isfinite(st).all()
            assert (st[:, 11] > 0.64).all() and (st[:, 11] < 0.67).all()
    assert seen_complex >= n // 2                             # the script really exercised the robot-contact kernels
    moved = st[:, 9] - x0[:, 0]
    assert (moved > 0.02).mean() > 0.9 and np.isfinite(st).all()


@pytest.mark.parametrize("use_ik,flags", [(0, 0), (1, 0), (0, _capi.F_FORCE_GENERAL), (0, _capi.F_COMPLEX_LANES)])
def test_action_repeat(panda, hip_lib, use_ik, flags):
    parity.check_action_repeat(_capi.Engine, hip_lib, panda["table"], use_ik=use_ik, flags=flags)


def test_force_limited_motors(hip_lib, panda):
    eng = parity.check_panda_force_limited(_capi.Engine, hip_lib, panda["table"], n=70)
    info = eng.kernel_info()
    assert info[2] == 1 and info[3] == 70, "the last step (from the unlimited arm's settled state) belongs to the lane-per-env kernel"


def test_world_check_contact(panda, hip_lib):
    """WorldEnv.check_contact on states downloaded from the GPU engine (host-side query, model/contacts.py) against the oracle's contact list"""
    from test_vec_env import check_world_contacts
    check_world_contacts(panda, hip_lib)


def test_scripted_push_closed_loop_against_oracle(panda, hip_lib):
    """Long horizon WITH pushing: the scripted push, closed loop in the engine (GPU) and in the oracle, free running for 280 steps; the
    cubes' final displacements compared per env."""
    rep = parity.check_panda_push_closed_loop(_capi.Engine, hip_lib, panda["table"], n=16)
    print("closed-loop push:", rep)
    assert rep["touched_envs"] == 16


@pytest.mark.parametrize("flags", [0, _capi.F_FORCE_GENERAL])
def test_round_objects(panda, hip_lib, flags):
    """sphere / cylinder stand-ins of the round objects (YcbTennisBall, the cans, duck_vhacd): reset, rolling without slipping, single
    steps and robot-object contacts against the oracle (k_fast + ObjStep / k_row_list, and the general row kernel)"""
    rep = parity.check_round_objects(_capi.Engine, hip_lib, panda["table"], names=("YcbTennisBall", "YcbTomatoSoupCan", "YcbMasterChefCan", "duck_vhacd", "YcbPear"), n=24, flags=flags)
    print("round objects:", {k: {kk: v[kk] for kk in ("travel_cm", "free_run_obj_pos_diff", "obj_w", "skipped", "compared", "robot_contact_compared") if kk in v} for k, v in rep.items()})


def test_convex_hull_objects(panda, hip_lib):
    """the object as a convex hull (pbre_set_object_hull; k_step: Core's hull candidates + sphere_hull) -- the cube as its 8
    vertices (reproduces the box primitive: object-table rows bit for bit), a tetrahedron, a 20- and a 32-vertex blob -- reset, sliding /
    spinning single steps and robot-object contacts at faces and vertices against the oracle's brute-force hull"""
    rep = parity.check_hull_objects(_capi.Engine, hip_lib, panda["table"], n=24)
    print("hull objects:", {k: {kk: v[kk] for kk in ("reset_rel", "rest_height", "obj_w", "skipped", "compared", "robot_contact_compared") if kk in v} for k, v in rep.items()})
    assert all(v["robot_contact_compared"] >= 12 for v in rep.values())


def test_closed_form_motor_rows_match_the_sequential_rows(panda, hip_lib):
    """The simple class applies its 150 sweeps over the clamp-free motor rows in closed form (a matrix power, Fast::motor_closed);
    PBRE_F_SEQ_MOTORS runs Bullet's sequential rows instead.  Same states, one step each: the object (which the motor rows do not
    touch) bit for bit, everything else within a quarter of the single-step bounds against the oracle."""
    n = 256
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib)
    c = _capi.Engine(panda["table"], **kw)
    s = _capi.Engine(panda["table"], flags=_capi.F_SEQ_MOTORS, **kw)
    c.reset(); s.reset()
    rng = np.random.default_rng(8)
    for _ in range(4):
        a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        s.set_state(c.get_state())
        (obc, rwc, dnc), (obs_, rws, dns) = c.step(a), s.step(a)
        sc, ss = c.get_state(), s.get_state()
        assert np.array_equal(sc[:, 9:16], ss[:, 9:16]) and np.array_equal(sc[:, 25:31], ss[:, 25:31]) and np.array_equal(dnc, dns)
        out_s = np.concatenate([obs_, rws[:, None], dns[:, None]], 1).astype(np.float64)
        q = parity.panda_quantities(sc, ss.astype(np.float64), obc, out_s, rwc)
        parity.assert_within(q, dict((k, 0.25 * v) for k, v in parity.TOL.items()), "(closed-form motor rows against the sequential rows)")


@pytest.mark.parametrize("use_ik,action_repeat", [(0, 1), (1, 1), (0, 2)])
def test_three_waves_per_simd_build_is_bit_identical(panda, hip_lib, monkeypatch, use_ik, action_repeat):
    """k_fast exists in two builds -- 256 VGPRs / two waves per SIMD and 168 VGPRs / three (spills in its setup phase) -- and launch_step
    picks per step; at the headline batch the 168-register build steps the stationary mix, at a test's batch sizes it is never picked.
    PBRE_FAST3=1 takes it whenever complex envs are reported: same states (contact-rich ones among them, so that the complex-env kernel
    runs beside it), several steps, results bit for bit those of the 256-register build (PBRE_FAST3=0)."""
    n = 4096
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=40,
              use_ik=use_ik, action_repeat=action_repeat)           # (joint control, IK control, the inner iterations of action_repeat: all k_fast modes)
    monkeypatch.setenv("PBRE_PAIR", "0")            # (the two builds of k_fast: not the pair kernel, whatever the module's mapping fixture says)
    monkeypatch.setenv("PBRE_FUSED", "0")           # (... and k_fast as a kernel of its own: the one-launch step has the 256-register build only)
    monkeypatch.setenv("PBRE_FAST3", "1")
    a = _capi.Engine(panda["table"], **kw)
    monkeypatch.setenv("PBRE_FAST3", "0")
    b = _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()
    # a few contact-rich states so that complex envs exist from the first step on
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 8, 8).astype(np.float32)
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(9)
    for _ in range(60):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    assert np.array_equal(a.get_state(), b.get_state())
    ia, ib = a.kernel_info(), b.kernel_info()
    assert ia[8] > 0 and ib[8] == 0, (ia, ib)          # steps whose k_fast was the three-waves-per-SIMD build
    assert ia[9] <= 168 and ia[0] > 168, ia            # its register count / the default build's


@pytest.mark.parametrize("use_ik,action_repeat", [(0, 1), (1, 1), (0, 2)])
def test_one_launch_step_is_bit_identical_to_the_two_kernel_step(panda, hip_lib, monkeypatch, simple_env_mapping, use_ik, action_repeat):
    """Round 5: env.step() is ONE launch (k_fused: the complex envs' row blocks and the simple envs' waves -- k_fast's or, under the pair
    mapping, k_fast_pair's, two pairs per block -- in one grid) instead of two kernels on two streams with fork / join events.  Same
    device functions: a full reset through each launch form, contact-rich states among the envs so that both parts of the grid have work,
    60 steps with auto-reset -- rows, states and classes bit for bit (PBRE_FUSED=0: the two-kernel step)."""
    n = 4096
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=40,
              use_ik=use_ik, action_repeat=action_repeat)
    monkeypatch.setenv("PBRE_FUSED", "1")
    a = _capi.Engine(panda["table"], **kw)
    monkeypatch.setenv("PBRE_FUSED", "0")
    b = _capi.Engine(panda["table"], **kw)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob) and np.array_equal(a.get_state(), b.get_state())
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 8, 8).astype(np.float32)
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(9)
    seen = 0
    for _ in range(60):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
        seen = max(seen, a.kernel_info()[5])
    assert np.array_equal(a.get_state(), b.get_state())
    ia, ib = a.kernel_info(), b.kernel_info()
    assert seen > 0, "no complex env: the row blocks of the fused grid had no work"
    # steps that were one launch (IK control: while most of the batch is complex -- the home pose's IK solution lies beyond a joint limit --
    # the lane-per-env complex kernel k_fast_rc steps them, on its own stream as before)
    assert (ia[13] >= 60 or use_ik) and ib[13] == 0, (ia, ib)
    assert 0 < ia[14] <= 256, ia
    assert (ia[10] > 0) == (simple_env_mapping == "pair") and (ib[10] > 0) == (simple_env_mapping == "pair"), (ia, ib)


def test_tail_pairs_of_the_one_launch_step_are_bit_identical(panda, hip_lib, monkeypatch):
    """Round 6: in k_fused's 64-thread grid the last chunks of a machine-filling batch -- the ones the row waves keep out of the first round --
    are stepped by a robot wave and an object wave (two one-wave blocks, the object's new pose through a global record + per-lane sequence
    words) instead of one k_fast wave.  Forced here at a small size (PBRE_TAIL_PAIR=n: always the last n chunks; PBRE_PAIR=0: the 64-thread
    grid): a full reset through each form, contact-rich states among the envs, 60 steps with auto-reset -- rows, states and classes bit for
    bit those without tail pairs, no env-step lost to a wait (guard counter 0), and the sequence numbers' wrap in between."""
    n = 4096
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=40)
    monkeypatch.setenv("PBRE_PAIR", "0")
    monkeypatch.setenv("PBRE_OBJV_SEQ0", str(0x7fffffff - 230))       # (201 settle launches of the reset, then the wrap within the 60 steps)
    monkeypatch.setenv("PBRE_TAIL_PAIR", "40")
    a = _capi.Engine(panda["table"], **kw)
    monkeypatch.setenv("PBRE_TAIL_PAIR", "0")
    b = _capi.Engine(panda["table"], **kw)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob) and np.array_equal(a.get_state(), b.get_state())
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 8, 8).astype(np.float32)
    st = a.get_state()
    st[n - len(S):, :S.shape[1]] = S              # (among the tail chunks' envs: lanes of complex envs idle in either mapping)
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(9)
    for _ in range(60):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    assert np.array_equal(a.get_state(), b.get_state())
    ia, ib = a.kernel_info(), b.kernel_info()
    assert ia[15] >= 60 and ib[15] == 0, (ia, ib)      # steps (since the reset) that had tail pairs
    assert ia[12] == 0 and ib[12] == 0, (ia, ib)       # NaN / Inf guard: no env-step lost to a wait that ran out


def test_one_launch_step_with_thousands_of_complex_envs(panda, hip_lib, monkeypatch, simple_env_mapping):
    """The row part of k_fused when every slot makes several trips over the complex lists: 4096 envs, ALL of them in contact-rich states
    (robot-table, robot-object, joints at limits: both complex lists), the row kernel forced (F_COMPLEX_ROWS) -- several hundred work items
    over at most 256 slots, so each slot's object wave hands over several rounds of twists (per-env global records + sequence number in
    the 64-thread grid, LDS + barriers in the 256-thread one).  Rows, states and classes bit for bit those of the two-kernel step."""
    n = 4096
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(3), 24, 24).astype(np.float32)
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET | _capi.F_COMPLEX_ROWS, max_steps=40)
    monkeypatch.setenv("PBRE_FUSED", "1")
    a = _capi.Engine(panda["table"], **kw)
    monkeypatch.setenv("PBRE_FUSED", "0")
    b = _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()
    st = a.get_state()
    reps = (n + len(S) - 1) // len(S)
    st[:, :S.shape[1]] = np.tile(S, (reps, 1))[:n]
    a.set_state(st); b.set_state(st)
    assert a.kernel_info()[5] > 3000, a.kernel_info()          # complex envs of the current state
    rng = np.random.default_rng(11)
    for k in range(12):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), "step %d" % k
    assert np.array_equal(a.get_state(), b.get_state())
    ia, ib = a.kernel_info(), b.kernel_info()
    assert ia[13] >= 12 and ib[13] == 0, (ia, ib)


def test_one_launch_step_across_the_wrap_of_its_sequence_numbers(panda, hip_lib, monkeypatch):
    """k_fused's 64-thread grid marks an object wave's side record complete with the launch's sequence number; after 2^31 - 1 launches the
    numbers start over behind a clear of the records.  PBRE_OBJV_SEQ0 starts them eight launches before that: 20 steps across the wrap,
    complex envs present, bit for bit the two-kernel step."""
    n = 2048
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(5), 12, 12).astype(np.float32)
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=40)
    monkeypatch.setenv("PBRE_PAIR", "0")                     # the 64-thread grid, whatever the module's mapping fixture says
    monkeypatch.setenv("PBRE_FUSED", "1")
    a = _capi.Engine(panda["table"], **kw)
    monkeypatch.setenv("PBRE_FUSED", "0")
    b = _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()                                     # (201 one-launch settle steps each: sequence numbers 1 .. 201)
    a.close()
    monkeypatch.setenv("PBRE_FUSED", "1")
    monkeypatch.setenv("PBRE_OBJV_SEQ0", str(2 ** 31 - 1 - 201 - 8))
    a = _capi.Engine(panda["table"], **kw)
    a.reset()
    st = b.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(13)
    for k in range(20):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y), "step %d" % k
    assert np.array_equal(a.get_state(), b.get_state())
    assert a.kernel_info()[13] >= 20


def test_staged_copies_match_zero_copy_host_buffers(panda, hip_lib, monkeypatch):
    """pbre_step with page-locked buffers: by default the kernels read the actions from and write the rows to host memory themselves
    (PBRE_ZERO_COPY=3); =0 stages them through device buffers with hipMemcpyAsync.  Same rows, bit for bit, complex envs included
    (their rows are written by the side stream's kernel)."""
    n = 512
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib)
    monkeypatch.setenv("PBRE_ZERO_COPY", "3")
    a = _capi.Engine(panda["table"], **kw)
    monkeypatch.setenv("PBRE_ZERO_COPY", "0")
    b = _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(2), 6, 6).astype(np.float32)
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(10)
    for _ in range(6):
        act = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        for x, y in zip(a.step(act), b.step(act)):
            assert np.array_equal(x, y)
    assert a.kernel_info()[7] > 0          # complex env-steps were among them


def test_pipelined_host_path_is_bit_equal_to_the_synchronous_one(panda, hip_lib):
    """pbre_step_async / pbre_step_wait (round 6: upload, kernels and download of consecutive steps overlap on three streams, two steps in
    flight) against pbre_step: the same rows, bit for bit, over 40 steps with auto-reset and contact-rich envs; the bookkeeping errors
    (a third step in flight, a wait without a step) are refused; a host-synchronous call in between does not lose rows."""
    n = 4096
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=30)
    a = _capi.Engine(panda["table"], **kw)
    b = _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(2), 6, 6).astype(np.float32)
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(12)
    acts = rng.uniform(-1, 1, (40, n, 7)).astype(np.float32)
    ref = [a.step(acts[t]) for t in range(40)]
    with pytest.raises(RuntimeError):
        b.step_async(acts[0]); b.step_async(acts[1]); b.step_async(acts[2])      # the third is refused (nothing was enqueued for it)
    got = [b.step_wait(copy=True), b.step_wait(copy=True)]
    with pytest.raises(RuntimeError):
        b.step_wait()
    b.step_async(acts[2])
    for t in range(3, 40):
        b.step_async(acts[t])
        if t == 20:
            assert np.isfinite(b.get_state()).all()           # a host-synchronous entry point with two steps in flight
        got.append(b.step_wait(copy=True))
    got.append(b.step_wait(copy=True))
    assert len(got) == 40
    for t in range(40):
        for x, y in zip(ref[t], got[t]):
            assert np.array_equal(x, y), t
    assert np.array_equal(a.get_state(), b.get_state()) and a.kernel_info()[7] > 0
    a.close(); b.close()


@pytest.mark.parametrize("use_ik", [0, 1])
def test_complex_env_results_do_not_depend_on_their_wave_mates(panda, hip_lib, use_ik):
    """Two identical engines side by side, contact-rich envs among 4096, 60 steps with auto-reset: bit-identical throughout.  The
    complex envs of a step are appended to their list by atomics, so which of them share a wave of the row kernel differs from run to
    run -- and Core::step picks its solver path per wave.  The paths are the same arithmetic; implicit FMA contraction made them differ
    in the last bit (round 3: found by tools/diag_fast3.py, fixed by `#pragma clang fp contract(off)` in pbre_core.hpp)."""
    n = 4096
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=40, use_ik=use_ik)
    a = _capi.Engine(panda["table"], **kw)
    b = _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 8, 8).astype(np.float32)
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    for trial in range(3):
        rng = np.random.default_rng(11 + trial)
        for _ in range(60):
            act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
            for x, y in zip(a.step(act), b.step(act)):
                assert np.array_equal(x, y)
        assert np.array_equal(a.get_state(), b.get_state())
    assert a.kernel_info()[7] > 0


def test_simple_env_results_do_not_depend_on_their_wave_mates(panda, hip_lib):
    """k_fast has a second copy of its solver loop for the usual wave (all four object-table slots in use in every lane) and a per-lane
    fallback for clamping motors; which copy a wave runs must not change a lane's result: one wave of 64 envs, in the second engine
    env 1 has a tilted cube (fewer table contacts) and env 2 a sliding one -- the other 62 envs bit-identical over 40 steps."""
    n = 64
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib)
    a, b = _capi.Engine(panda["table"], **kw), _capi.Engine(panda["table"], **kw)
    a.reset(); b.reset()
    st = a.get_state()
    sb = st.copy()
    ang = np.deg2rad(20.0)
    sb[1, 12:16] = [np.sin(ang / 2), 0, 0, np.cos(ang / 2)]
    sb[1, 11] += 0.01
    sb[2, 25:28] = [0.3, -0.2, 0.0]
    a.set_state(st); b.set_state(sb)
    others = np.ones(n, bool); others[[1, 2]] = False
    rng = np.random.default_rng(0)
    for _ in range(40):
        act = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        (oa, ra, da), (ob, rb, db) = a.step(act), b.step(act)
        assert np.array_equal(oa[others], ob[others]) and np.array_equal(ra[others], rb[others])
        assert np.array_equal(a.get_state()[others], b.get_state()[others])
    assert not np.array_equal(a.get_state()[1], b.get_state()[1])


def test_sliding_cube_stops_where_coulomb_friction_says(panda, hip_lib):
    """analytic contact KAT through the engine (parity.check_sliding_cube_kat): a cube given a horizontal velocity stops after the
    distance Coulomb friction predicts -- along a pyramid axis and, sqrt(2) harder, along the diagonal -- straight, without turning"""
    eng = _capi.Engine(panda["table"], task=1, num_envs=1, obj_pose_rnd_std=0.0, tg_pose_rnd_std=0.0, lib=hip_lib)
    eng.reset()
    st = eng.get_state()
    ph = eng.get_physics()
    zero = np.zeros((1, 7), np.float32)

    def step(s):
        eng.set_state(np.asarray(s, np.float32))
        eng.step(zero)
        return eng.get_state()
    rep = parity.check_sliding_cube_kat(step, st, {"mu": ph.obj_mu * ph.table_mu, "g": -ph.gravity_z, "kl": ph.lin_damping, "dt": ph.dt})
    print("sliding cube:", rep)
    print("free fall:", parity.check_free_fall_kat(step, st, {"g": -ph.gravity_z, "kl": ph.lin_damping, "dt": ph.dt}))


def test_sliding_ball_and_can_end_up_rolling_at_the_analytic_speed(panda, hip_lib):
    """analytic KAT for the round primitives through the engine (parity.check_rolling_onset_kat): 5/7 v0 for the ball, 2/3 v0 for the
    lying can, whatever the friction coefficient"""
    from pybullet_robot_envs.model.objects import object_physics
    keep = []

    def make(name):
        ph = object_physics(name)
        eng = _capi.Engine(panda["table"], task=1, num_envs=1, obj_pose_rnd_std=0.0, tg_pose_rnd_std=0.0, lib=hip_lib, phys=ph)
        keep.append(eng)
        eng.reset()
        zero = np.zeros((1, 7), np.float32)

        def step(s):
            eng.set_state(np.asarray(s, np.float32))
            eng.step(zero)
            return eng.get_state()
        return eng.get_state(), ph["obj_h"][0], step
    print("rolling onset:", parity.check_rolling_onset_kat(make))



@pytest.mark.parametrize("use_ik,obj", [(0, None), (1, None), (0, "YcbTennisBall"), (0, "YcbMustardBottle")])
def test_pair_kernel_is_bit_identical_to_k_fast(panda, hip_lib, monkeypatch, use_ik, obj):
    """k_fast_pair (the small-batch mapping: two waves per 64 envs) against k_fast on the same states -- a full reset through each, contact-rich
    envs among them (k_row_list beside it), 60 steps with auto-reset, per-env object parameters: rows, states and classes bit for bit.
    That is what keeps the sharding invariance bitwise although shard sizes select different kernels."""
    phys = None
    if obj:
        from pybullet_robot_envs.model.objects import object_physics
        phys = object_physics(obj)
    ia, ib = parity.check_pair_split_is_bit_identical(_capi.Engine, hip_lib, panda["table"], panda, monkeypatch.setenv, n=2048, steps=60,
                                                      use_ik=use_ik, phys=phys)
    assert 0 < ia[11] <= 256, ia          # the pair kernel's register count
    print("steps taken by the pair kernel:", ia[10])


def test_pair_kernel_is_the_default_of_small_batches(panda, hip_lib, monkeypatch):
    """launch_step's rule (PBRE_PAIR unset): the pair kernel while its waves fit two per SIMD (<= 65536 envs on an MI355X)."""
    monkeypatch.delenv("PBRE_PAIR", raising=False)
    for n, want in ((4096, True), (65536, True), (131072, False)):
        eng = _capi.Engine(panda["table"], task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib)
        eng.reset()
        eng.step(np.zeros((n, 7), np.float32))
        info = eng.kernel_info()
        assert (info[10] > 0) == want, (n, info)
        eng.close()


@pytest.mark.parametrize("flags", [0, _capi.F_COMPLEX_ROWS, _capi.F_COMPLEX_LANES, _capi.F_FORCE_GENERAL])
def test_nan_inf_guard(panda, hip_lib, flags):
    """SURVEY section 5 (failure detection): NaN / Inf injected into the state of five envs -- counted (pbre_kernel_info[12]), returned
    as reward 0 / done 1, restarted under PBRE_F_AUTO_RESET; every other env bit-unchanged.  flags: which kernels see them (k_fast /
    k_fast_pair + k_row_list, k_fast_rc, the general 16-lane kernel)."""
    print(parity.check_nan_guard(_capi.Engine, hip_lib, panda["table"], panda, flags_extra=flags))


@pytest.mark.parametrize("flags", [0, _capi.F_COMPLEX_ROWS, _capi.F_COMPLEX_LANES, _capi.F_FORCE_GENERAL])
def test_four_robot_object_contact_slots(panda, hip_lib, flags):
    """four robot-object slots -- the cube pinched between the fingers (both spheres of both fingers, sometimes the palm as a fifth
    candidate), one step against the four-slot oracle; k_row_list, k_fast_rc and the general kernel"""
    rep = parity.check_four_robot_object_slots(_capi.Engine, hip_lib, panda["table"], panda, flags=flags)
    print({k: v for k, v in rep.items() if k != "robot_object_contacts_per_state"})


@pytest.mark.parametrize("use_ik", [0, 1])
def test_solver_residual_threshold_free_space(panda, hip_lib, use_ik):
    """pbre_physics.solver_residual_threshold = 1e-7 on the device (k_fast<MODE, WPS, RT>): reset, then single steps against the oracle
    with the same threshold; per-env sweep counts (pbre_get_sweeps) equal the oracle's except for a bounded fraction of one-sweep flips."""
    rep = parity.check_residual_threshold(_capi.Engine, hip_lib, panda["table"], n=192, steps=3, use_ik=use_ik)
    assert rep["early"] >= rep["compared"] // 2


@pytest.mark.parametrize("flags", [0, _capi.F_COMPLEX_ROWS, _capi.F_COMPLEX_LANES, _capi.F_FORCE_GENERAL])
def test_solver_residual_threshold_contact_rich_states(panda, hip_lib, flags):
    """... crafted contact-rich states on k_row_list<MODE, RT> (default for few complex envs; F_COMPLEX_ROWS pins it), k_fast_rc<MODE, RT>
    (F_COMPLEX_LANES) and k_step<MODE, RT> (F_FORCE_GENERAL)."""
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 24, 24)
    parity.check_residual_threshold(_capi.Engine, hip_lib, panda["table"], states=S, steps=1, flags=flags, tol=parity.TOL_CONTACT, skip_ambiguous=True)


def test_solver_residual_threshold_moving_cubes_are_complex_class_states(panda, hip_lib):
    rep = parity.check_residual_threshold_moving_cubes(_capi.Engine, hip_lib, panda["table"], n=96)
    print({k: v for k, v in rep.items() if k not in ("worst", "worst_flip")})


def test_solver_residual_threshold_results_do_not_depend_on_wave_mates_or_sharding(panda, hip_lib):
    """With the threshold on an env leaves the sweep loop on its own while its wave goes on: two engines side by side (complex envs land in
    different row-kernel waves from run to run) and a 2-shard split of the same batch are bit-identical over 40 steps with auto-reset."""
    n = 4096
    kw = dict(task=1, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=hip_lib, flags=_capi.F_AUTO_RESET, max_steps=30)
    a = _capi.Engine(panda["table"], num_envs=n, **kw)
    b = _capi.Engine(panda["table"], num_envs=n, **kw)
    h0 = _capi.Engine(panda["table"], num_envs=n // 2, env_id_base=0, **kw)
    h1 = _capi.Engine(panda["table"], num_envs=n // 2, env_id_base=n // 2, **kw)
    _, ora = parity.make_pair(_capi.Engine, hip_lib, panda["table"], 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 8, 8).astype(np.float32)
    for e in (a, b, h0, h1):
        e.reset()
        e.set_physics(solver_residual_threshold=1e-7)
    st = a.get_state()
    st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st); h0.set_state(st[:n // 2]); h1.set_state(st[n // 2:])
    rng = np.random.default_rng(12)
    early = 0
    for _ in range(40):
        act = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
        ra, rb, r0, r1 = a.step(act), b.step(act), h0.step(act[:n // 2]), h1.step(act[n // 2:])
        for x, y, z0, z1 in zip(ra, rb, r0, r1):
            assert np.array_equal(x, y) and np.array_equal(x, np.concatenate([z0, z1]))
        sw = a.get_sweeps()
        assert np.array_equal(sw, b.get_sweeps()) and np.array_equal(sw, np.concatenate([h0.get_sweeps(), h1.get_sweeps()]))
        early += int((sw < 150).sum())
    assert np.array_equal(a.get_state(), b.get_state())
    assert a.kernel_info()[7] > 0 and early > 20 * n


def test_closed_form_object_rows_match_the_sequential_rows(panda, hip_lib):
    """k_fast / k_fast_pair with the object block's closed form against the same kernels with PBRE_F_SEQ_OBJECT (all 150 sweeps row by
    row), on cubes at rest, sliding, spinning, dropped, tilted, tumbling, pressed (parity.check_closed_form_object_rows); the bound behind
    the closed form is checked independently in tests/test_objblock_bound.py."""
    rep = parity.check_closed_form_object_rows(_capi.Engine, hip_lib, panda["table"], n=4096, steps=12)
    assert rep["bitwise_equal_env_steps"] > 0       # lanes that failed the bound ran the explicit rows
