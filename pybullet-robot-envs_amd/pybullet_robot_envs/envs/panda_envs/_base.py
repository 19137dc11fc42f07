"""Shared implementation of the three Panda task envs (reach / push / push-goal).

Each public class keeps the reference constructor signature and Gym surface
(reset() -> obs, step(a) -> (obs, reward, done, info), render(), seed(), observation_space,
action_space, _robot, _world, _env_step_counter, terminated) and adds optional trailing kwargs
`num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False` (auto_reset: a finished env is re-initialised
inside the step that finished it -- the step returns that transition's reward/done with the first observation of the
next episode, Isaac-Gym style -- using the snapshot reset of DESIGN.md section 6).  With num_envs == 1 the return shapes are the
reference's ((obs_dim,) float64, 0-d reward, 0-d float32 done, {}); with num_envs = N everything is
stacked [N, ...].  All per-step work is one fused HIP kernel behind the C-ABI (include/pbre.h)."""
import numpy as np

from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs._gym import Env, spaces, seeding
from pybullet_robot_envs.envs.panda_envs.panda_env import pandaEnv
from pybullet_robot_envs.envs.world_envs.world_env import WorldEnv
from pybullet_robot_envs.envs.utils import goal_distance, scale_gym_data


class PandaTaskBase(Env):
    metadata = {'render.modes': ['human', 'rgb_array'],
                'video.frames_per_second': 50}
    _TASK = _capi.TASK_PUSH

    def _setup(self, numControlledJoints, use_IK, action_repeat, obj_name, renders, max_steps, obj_pose_rnd_std,
               tg_pose_rnd_std, includeVelObs, target_dist_min, num_envs, device_id, env_id_base, seed, _lib, auto_reset=False):
        self._timeStep = 1. / 240.
        self.action_dim = []
        self._use_IK = use_IK
        self._action_repeat = action_repeat
        self._observation = []
        self._renders = renders        # accepted for API parity; there is no GUI (render() returns an empty array)
        self._max_steps = max_steps
        self._target_dist_min = target_dist_min
        self._tg_pose_rnd_std = tg_pose_rnd_std
        self._obj_pose_rnd_std = obj_pose_rnd_std
        self.includeVelObs = includeVelObs
        self.num_envs = int(num_envs)
        self._auto_reset = bool(auto_reset)

        # "connect": one engine session instead of one PyBullet client (panda_push_gym_env.py:56-62)
        self._physics_client_id = _client.connect(num_envs, device_id, env_id_base, seed, _lib)
        self._client = _client.get(self._physics_client_id)

        # Load robot (panda_push_gym_env.py:65)
        self._robot = pandaEnv(self._physics_client_id, use_IK=self._use_IK, joint_action_space=numControlledJoints)

        # Load world environment (:68-70)
        self._world = WorldEnv(self._physics_client_id, obj_name=obj_name, obj_pose_rnd_std=obj_pose_rnd_std,
                               workspace_lim=self._robot.get_workspace())

        # limit robot workspace to table plane (:73-75 push: h - 0.2; panda_reach_gym_env.py:68-70: h)
        workspace = self._robot.get_workspace()
        workspace[2][0] = self._world.get_table_height() - (0.0 if self._TASK == _capi.TASK_REACH else 0.2)
        self._robot.set_workspace(workspace)

        self._build_engine()

        # Define spaces
        self.observation_space, self.action_space = self.create_gym_spaces()
        self.seed()
        # self.reset()  (the reference does not reset in the constructor either)

    # ------------------------------------------------------------------ engine
    def _build_engine(self):
        c = self._client
        if c.engine is not None:
            c.engine.close()
        ws = self._world.get_workspace()
        cfg = _capi.Config()
        overrides = dict(device_id=c.device_id, env_id_base=c.env_id_base, seed=c.seed, action_repeat=int(self._action_repeat),
                         num_controlled_joints=self._robot.joint_action_space, max_steps=int(self._max_steps),
                         obj_pose_rnd_std=float(self._obj_pose_rnd_std), tg_pose_rnd_std=float(self._tg_pose_rnd_std),
                         target_dist_min=float(self._target_dist_min), h_table=float(self._world.get_table_height()),
                         flags=_capi.F_AUTO_RESET if self._auto_reset else 0, use_ik=1 if self._use_IK else 0)
        self._robot._own_engine = False      # the engine of this client is the task env's from now on (pandaEnv._robot_level)
        c.engine = _capi.make_engine(self._robot.robot_table, devices=c.devices, task=self._TASK, num_envs=c.num_envs, lib=c.lib,
                                      phys=self._world.object_physics(), **overrides)
        rws = self._robot.get_workspace()
        for a in range(3):
            for b in range(2):
                assert abs(c.engine.cfg.ws_lim[a][b] - ws[a][b]) < 1e-12, "workspace differs from the engine default"
                assert abs(c.engine.cfg.robot_ws[a][b] - rws[a][b]) < 1e-12, "robot workspace differs from the engine default"
        assert c.engine.act_dim == self._robot.get_action_dim()
        self._engine = c.engine

    def close(self):
        _client.disconnect(self._physics_client_id)

    # ------------------------------------------------------------------ spaces
    def create_gym_spaces(self):
        obs, obs_lim = self.get_extended_observation()
        observation_low = [el[0] for el in obs_lim]
        observation_high = [el[1] for el in obs_lim]
        observation_space = spaces.Box(np.array(observation_low), np.array(observation_high), dtype='float32')
        self.action_dim = self._robot.get_action_dim()
        action_high = np.array([1] * self.action_dim)
        action_space = spaces.Box(-action_high, action_high, dtype='float32')
        return observation_space, action_space

    def _squeeze(self, x):
        return x[0] if self.num_envs == 1 else x

    # ------------------------------------------------------------------ gym API
    def reset(self, mask=None, snapshot=False):
        """reset_simulation + target sampling + observation (panda_push_gym_env.py:105-148), on the GPU for all
        envs (or those selected by `mask`, a batched-only extension).  snapshot=True (with a mask): the selected envs restart from
        the settled snapshot of the last full reset (pbre_reset_snapshot: one kernel instead of 201 settle launches, within 2e-5 /
        5e-5 of the explicit reset) -- what a vectorised rollout loop wants for the envs that just finished."""
        if snapshot and mask is not None:
            raw = self._engine.reset_snapshot(mask).astype(np.float64)
        else:
            raw = self._engine.reset(mask).astype(np.float64)
        return self._squeeze(scale_gym_data(self.observation_space, raw))

    def get_extended_observation(self):
        raw = self._engine.observe().astype(np.float64)
        lo, hi = self._engine.obs_limits()
        lim = [[float(a), float(b)] for a, b in zip(lo.astype(np.float64), hi.astype(np.float64))]
        lim = self._exact_limits(lim)
        self._observation = self._squeeze(raw)
        return np.array(self._observation), lim

    def _exact_limits(self, lim32):
        # the reference builds the limits as Python floats; rebuild them from the same sources so that
        # np.array(low) -> float32 rounds identically (create_gym_spaces, panda_push_gym_env.py:83-103)
        import math as m
        lim = []
        lim.extend(self._robot.get_observation_limits())
        wl = self._world.get_workspace()
        lim.extend(wl)
        lim.extend([[-m.pi, m.pi]] * 3)
        lim.extend([[-0.5, 0.5]] * 3)
        lim.extend([[0, 2 * m.pi]] * 3)
        if self._TASK != _capi.TASK_REACH:
            lim.extend(wl[:3])
        assert len(lim) == len(lim32)
        assert np.allclose(np.array(lim, dtype=np.float32), np.array(lim32, dtype=np.float32), atol=0, rtol=0), \
            "engine observation limits differ from the Python-side limits"
        return lim

    def apply_action(self, action):
        """The action half of the reference's step() (panda_push_gym_env.py:191-255, icub_reach_gym_env.py:204-259): scaled action ->
        motors (through the IK when use_IK), then action_repeat x (stepSimulation, step counter, termination check).  Here that is the
        fused GPU step itself; what it returns is dropped, and get_extended_observation() / _termination() / _compute_reward() called
        afterwards -- the rest of the reference's step() -- read the state it left (without auto_reset: a finished env that restarted
        inside the step shows its new episode)."""
        self._raw_step(action)

    def _raw_step(self, action):
        a = np.asarray(action, dtype=np.float32)
        if a.ndim == 1:
            a = a[None]
        return self._engine.step(a)

    def step(self, action):
        raw, reward, done = self._raw_step(action)
        scaled_obs = scale_gym_data(self.observation_space, raw.astype(np.float64))
        if self.num_envs == 1:
            return scaled_obs[0], np.array(np.float64(reward[0])), np.array(np.float32(done[0])), {}
        return scaled_obs, reward.astype(np.float64), done.astype(np.float32), {}

    # ------------------------------------------------------------------ device-resident API (torch tensors, no host hop)
    def step_tensor(self, actions):
        """Batched step with device-resident data (replaces the DummyVecEnv hop of the reference's training scripts,
        train_ddpg_reaching.py:96): `actions` is a CUDA float32 tensor [num_envs, act_dim] on this env's GPU; returns
        (scaled_obs [N, obs_dim], reward [N], done [N]) CUDA float32 tensors that the caller owns (fresh tensors every call).
        Asynchronous and in stream order on torch's CURRENT stream (the kernels are enqueued on that very stream, so they see the
        actions the preceding torch kernels produce and later torch ops see the rows); observation scaling (scale_gym_data with
        the float32 Box limits) runs on the device in float32."""
        import torch
        a = actions.contiguous()
        assert a.is_cuda and a.dtype == torch.float32 and tuple(a.shape) == (self.num_envs, self._engine.act_dim)
        if getattr(self, "_t_out", None) is None or self._t_out.device != a.device:
            self._t_out = torch.empty((self.num_envs, self._engine.obs_dim + 2), device=a.device, dtype=torch.float32)
            box = self.observation_space["observation"] if hasattr(self.observation_space, "spaces") else self.observation_space
            self._t_low = torch.as_tensor(box.low, device=a.device)
            self._t_inv = 1.0 / (torch.as_tensor(box.high, device=a.device) - self._t_low)
        self._engine.step_device(a.data_ptr(), self._t_out.data_ptr(), _capi.torch_stream(a.device))
        od = self._engine.obs_dim
        obs = 2.0 * ((self._t_out[:, :od] - self._t_low) * self._t_inv) - 1.0
        return obs, self._t_out[:, od].clone(), self._t_out[:, od + 1].clone()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        self._world.seed(seed)
        self._robot.seed(seed)
        return [seed]

    def render(self, mode="rgb_array"):
        # no camera in the batched engine (the reference's own render() is broken: it reads an unset self._p,
        # panda_push_gym_env.py:267)
        return np.array([])

    # ------------------------------------------------------------------ reference attributes
    @property
    def _env_step_counter(self):
        return self._squeeze(self._engine.get_state_cols(self._engine.x_off + 3)[:, 0].astype(np.int64))

    @property
    def terminated(self):
        return self._squeeze(self._engine.get_state_cols(self._engine.x_off + 4)[:, 0].astype(np.int64))

    @property
    def _hand_pose(self):
        """Commanded hand pose (x, y, z, roll, pitch, yaw) of the IK mode (panda_push_gym_env.py:142-143, 197-222)."""
        return self._squeeze(self._engine.get_state_cols(self._engine.x_off + 6, 6).astype(np.float64))

    @property
    def _target_pose(self):
        return self._squeeze(self._engine.get_state_cols(self._engine.x_off, 3).astype(np.float64))

    def debug_gui(self):
        pass
