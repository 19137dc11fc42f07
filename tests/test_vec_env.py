"""BatchedVecEnv: the stable-baselines VecEnv protocol over one batched env (CPU lane emulation)."""
import numpy as np

from pybullet_robot_envs.envs import pandaPushGymEnv, pandaPushGymGoalEnv, iCubReachGymEnv
from pybullet_robot_envs.vec import BatchedVecEnv


def test_vec_env_masked_reset_semantics(emu_lib):
    env = pandaPushGymEnv(max_steps=2, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, num_envs=3, _lib=emu_lib)
    v = BatchedVecEnv(env)
    obs = v.reset()
    assert obs.shape == (3, 33) and v.num_envs == 3
    rng = np.random.default_rng(0)
    dones = []
    for t in range(4):
        v.step_async(rng.uniform(-1, 1, (3, 7)))
        obs, rew, done, infos = v.step_wait()
        assert obs.shape == (3, 33) and rew.shape == (3,) and done.dtype == bool and len(infos) == 3
        dones.append(done.copy())
        if done.any():
            i = int(np.nonzero(done)[0][0])
            assert "terminal_observation" in infos[i]
            assert int(np.atleast_1d(env._env_step_counter)[i]) == 0        # the finished env was reset
    assert np.array(dones).any()


def test_vec_env_device_reset_and_goal_env(emu_lib):
    env = pandaPushGymEnv(max_steps=2, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, num_envs=2, auto_reset=True, _lib=emu_lib)
    v = BatchedVecEnv(env)
    v.reset()
    for t in range(4):
        obs, rew, done, infos = v.step(np.zeros((2, 7)))
    assert obs.shape == (2, 33)
    g = BatchedVecEnv(pandaPushGymGoalEnv(max_steps=3, tg_pose_rnd_std=0.2, num_envs=2, _lib=emu_lib))
    o = g.reset()
    assert set(o) == {"observation", "achieved_goal", "desired_goal"} and o["observation"].shape == (2, 33)
    o, r, d, infos = g.step(np.zeros((2, 7)))
    assert "is_success" in infos[0] and r.shape == (2,)


def test_vec_env_single_icub(emu_lib):
    v = BatchedVecEnv(iCubReachGymEnv(max_steps=1, auto_reset=True, _lib=emu_lib))
    o = v.reset()
    assert o.shape == (1, 31)
    seen = False
    for t in range(4):
        o, r, d, infos = v.step(np.zeros((1, 3)))
        assert o.shape == (1, 31) and r.shape == (1,) and d.shape == (1,)
        if d[0]:
            seen = True
            assert int(v.env._env_step_counter) == 0           # finished in this step and already re-initialised on the "device"
    assert seen


def test_devices_kwarg_shards_the_batch_in_one_process(emu_lib):
    """`devices=[...]` on the Gym classes: one engine per listed device, env-id-keyed RNG -> the sharded env returns bit
    for bit what a single-engine env returns (here two shards on the emulation's single "device")."""
    from pybullet_robot_envs.envs import pandaPushGymEnv
    kw = dict(_lib=emu_lib, num_envs=6, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, max_steps=3)
    one, two = pandaPushGymEnv(**kw), pandaPushGymEnv(devices=[0, 0], **kw)
    assert type(two._engine).__name__ == "MultiEngine" and len(two._engine.shards) == 2 and two._engine.shards[1].cfg.env_id_base == 3
    assert np.array_equal(one.reset(), two.reset())
    rng = np.random.default_rng(0)
    for _ in range(5):
        a = rng.uniform(-1, 1, (6, 7))
        r1, r2 = one.step(a), two.step(a)
        assert all(np.array_equal(x, y) for x, y in zip(r1[:3], r2[:3]))
    assert np.array_equal(one._env_step_counter, two._env_step_counter)
    m = np.array([0, 1, 0, 0, 1, 0], np.uint8)
    assert np.array_equal(one.reset(mask=m), two.reset(mask=m))
    two.change_physics_params([0.1, 0.2, 0.3, 0.1, 0.2, 0.3], 0.7, 0.1, 0.04)
    assert np.allclose(two._engine.get_state_cols(44)[:, 0], [0.1, 0.2, 0.3, 0.1, 0.2, 0.3])
    one.close(); two.close()


def test_apply_action_is_the_action_half_of_step(emu_lib):
    """The reference's step() = apply_action + get_extended_observation + _termination + _compute_reward (panda_push_gym_env.py:230-259,
    icub_reach_gym_env.py:232-259): the same four calls on one env give what step() gives on its twin."""
    from pybullet_robot_envs.envs.utils import scale_gym_data
    for cls, kw, adim in ((pandaPushGymEnv, dict(obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2), 7), (iCubReachGymEnv, dict(use_IK=1), 3)):
        a_env = cls(num_envs=2, seed=7, _lib=emu_lib, **kw)
        b_env = cls(num_envs=2, seed=7, _lib=emu_lib, **kw)
        a_env.reset(); b_env.reset()
        rng = np.random.default_rng(1)
        for t in range(3):
            act = rng.uniform(-1, 1, (2, adim))
            obs, rew, done, _ = a_env.step(act)
            b_env.apply_action(act)
            raw, lim = b_env.get_extended_observation()
            assert np.allclose(scale_gym_data(b_env.observation_space, raw), obs, atol=1e-6)
            assert np.array_equal(b_env._termination(), done)
            assert np.allclose(b_env._compute_reward(), rew, rtol=2e-5, atol=2e-5)
        a_env.close(); b_env.close()


def test_world_check_contact_matches_the_oracle_contact_list(panda, emu_lib):
    check_world_contacts(panda, emu_lib)


def check_world_contacts(panda, lib):
    """WorldEnv.check_contact(body_id) (reference world_env.py:128-134) on contact-rich states: object-table and robot-object contact
    per env as in the contact list the oracle's stepSimulation builds for the same state (types 0 / 1 of orc_step_info)."""
    import orc
    import parity
    n = 24
    env = pandaPushGymEnv(num_envs=n, _lib=lib)
    env.reset()
    ora = orc.Oracle(panda["table"], task=1)
    eng = env._engine
    base = eng.get_state()[0].astype(np.float64)
    rng = np.random.default_rng(5)
    st = parity.contact_states(ora, panda, base, rng, n_table=8, n_obj=16).astype(np.float32)
    st[3, 11] += 0.3                     # one object in the air: no object-table contact
    eng.set_state(st)
    tab = env._world.check_contact(env._world.table_id)
    rob = env._world.check_contact(env._robot.robot_id)
    assert tab.shape == (n,) and rob.shape == (n,)
    home = np.array([0.0, -0.54, 0.0, -2.6, -0.30, 2.0, 1.0, 0.02, 0.02])
    exp_tab, exp_rob = np.zeros(n, bool), np.zeros(n, bool)
    for e in range(n):
        _, info = ora.sim_step(st[e].astype(np.float64), home, np.full(9, 0.2), np.ones(9))
        types = [info.type[c] for c in range(info.ncontacts)]
        exp_tab[e], exp_rob[e] = 0 in types, 1 in types
    assert not tab[3] and tab[:8].sum() == 7 and rob.sum() >= 8 and exp_tab.any() and exp_rob.any()
    assert np.array_equal(tab, exp_tab), (tab, exp_tab)
    assert np.array_equal(rob, exp_rob), (rob, exp_rob)
    single = pandaPushGymEnv(num_envs=1, _lib=lib)
    single.reset()
    assert single._world.check_contact(single._world.table_id) is True and single._world.check_contact(single._robot.robot_id) is False
    env.close(); single.close()


def test_scene_change_invalidates_the_settled_snapshot(emu_lib):
    """Round-2 advice: after a change of the scene (WorldEnv.load_object / set_physics with new geometry) the settled snapshot of the
    last full reset describes another scene.  The engine refuses snapshot restarts until a full reset re-records it -- mass / friction
    changes (domain randomisation) keep it -- and BatchedVecEnv falls back to the explicit masked reset."""
    import pytest
    env = pandaPushGymEnv(max_steps=2, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, num_envs=3, _lib=emu_lib)
    v = BatchedVecEnv(env)
    v.reset()
    eng = env._engine
    mask = np.array([1, 0, 0], np.uint8)
    eng.reset_snapshot(mask)                                     # valid after the full reset
    eng.set_physics(obj_mass=0.2, obj_mu=0.7)                    # domain randomisation: the rest pose does not change
    eng.reset_snapshot(mask)
    env._world.load_object("YcbMustardBottle")                   # a taller box: other rest height
    with pytest.raises(RuntimeError, match="stale"):
        eng.reset_snapshot(mask)
    rng = np.random.default_rng(0)
    for t in range(3):                                           # the adapter resets finished envs explicitly meanwhile
        obs, rew, done, infos = v.step(rng.uniform(-1, 1, (3, 7)))
    assert done.any() or True
    h = eng.get_physics().obj_h[2]
    z = eng.get_state()[:, 11]
    top = eng.get_physics().table_c[2] + eng.get_physics().table_h[2]
    assert np.all(np.abs(z - (top + h)) < 5e-3), (z, top + h)   # restarted envs rest ON the table with the new half height
    v.reset()
    eng.reset_snapshot(mask)                                     # a full reset re-records the snapshot
    a = pandaPushGymEnv(max_steps=2, num_envs=2, auto_reset=True, _lib=emu_lib)
    a.reset()
    a._world.load_object("YcbMustardBottle")
    with pytest.raises(RuntimeError, match="stale"):             # the in-kernel auto-reset must not restart from the old scene either
        a.step(np.zeros((2, 7)))
    a.reset()
    a.step(np.zeros((2, 7)))
