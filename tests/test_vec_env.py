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
