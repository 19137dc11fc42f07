"""iCubPushGymGoalEnv (reference pybullet_robot_envs/envs/icub_envs/icub_push_gym_goal_env.py): dict observation for HER,
sparse reward -(d > threshold), done = step budget or success."""
import numpy as np

from pybullet_robot_envs import _capi
from pybullet_robot_envs._gym import GoalEnv, spaces
from pybullet_robot_envs.envs.icub_envs.icub_push_gym_env import iCubPushGymEnv
from pybullet_robot_envs.envs.world_envs.world_env import get_objects_list
from pybullet_robot_envs.envs.utils import goal_distance, scale_gym_data


class iCubPushGymGoalEnv(GoalEnv, iCubPushGymEnv):
    _TASK = _capi.TASK_PUSH_GOAL

    def __init__(self,
                 action_repeat=1,
                 use_IK=1,
                 control_arm='l',
                 control_orientation=0,
                 obj_name=get_objects_list()[1],
                 obj_pose_rnd_std=0,
                 tg_pose_rnd_std=0.2,
                 renders=False,
                 max_steps=2000,
                 reward_type=1,
                 num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False, _lib=None, devices=None):
        device_id = devices if devices is not None else device_id
        iCubPushGymEnv.__init__(self, action_repeat, use_IK, control_arm, control_orientation, obj_name, obj_pose_rnd_std,
                                tg_pose_rnd_std, renders, max_steps, reward_type,
                                num_envs=num_envs, device_id=device_id, env_id_base=env_id_base, seed=seed, auto_reset=auto_reset, _lib=_lib)

    def create_gym_spaces(self):
        box, action_space = iCubPushGymEnv.create_gym_spaces(self)
        observation_space = spaces.Dict(dict(
            desired_goal=spaces.Box(-10, 10, shape=(3,), dtype='float32'),
            achieved_goal=spaces.Box(-10, 10, shape=(3,), dtype='float32'),
            observation=box,
        ))
        return observation_space, action_space

    def _split(self, raw):
        o = self._robot.get_observation_dim()          # robot observation, then world observation (object position first)
        return raw[:, o:o + 3].copy(), raw[:, -3:].copy()

    def _goal_dict(self, raw):
        raw = raw.astype(np.float64)
        ach, des = self._split(raw)
        obs = {'observation': scale_gym_data(self.observation_space['observation'], raw), 'achieved_goal': ach, 'desired_goal': des}
        if self.num_envs == 1:
            obs = dict((k, v[0]) for k, v in obs.items())
        return obs

    def get_goal_observation(self):
        raw = self._engine.observe().astype(np.float64)
        ach, des = self._split(raw)
        d = {'observation': raw, 'achieved_goal': ach, 'desired_goal': des}
        if self.num_envs == 1:
            d = dict((k, v[0]) for k, v in d.items())
        return d

    def reset(self, mask=None, snapshot=False):
        if snapshot and mask is not None:
            return self._goal_dict(self._engine.reset_snapshot(mask))
        return self._goal_dict(self._engine.reset(mask))

    def step(self, action):
        raw, reward, done = self._raw_step(action)
        obs = self._goal_dict(raw)
        succ = self._is_success(obs['achieved_goal'], obs['desired_goal'])
        info = {'is_success': succ}
        if self.num_envs == 1:
            return obs, np.float32(reward[0]), bool(done[0]), info
        return obs, reward.astype(np.float32), done.astype(bool), info

    def _termination(self):
        st = self._engine.get_state()
        return self._squeeze((st[:, self._engine.x_off + 3] > self._max_steps).astype(np.float32))

    def _is_success(self, achieved_goal, goal):
        d = goal_distance(np.asarray(achieved_goal)[..., :3], np.asarray(goal)[..., :3])
        return d <= self._target_dist_min

    def compute_reward(self, achieved_goal, goal, info):
        d = goal_distance(np.asarray(achieved_goal)[..., :3], np.asarray(goal)[..., :3])
        return -(d > self._target_dist_min).astype(np.float32)
