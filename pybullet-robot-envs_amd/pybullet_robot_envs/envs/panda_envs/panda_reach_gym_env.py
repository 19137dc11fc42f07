"""pandaReachGymEnv (reference pybullet_robot_envs/envs/panda_envs/panda_reach_gym_env.py)."""
import numpy as np

from pybullet_robot_envs import _capi
from pybullet_robot_envs.envs.panda_envs._base import PandaTaskBase
from pybullet_robot_envs.envs.world_envs.world_env import get_objects_list
from pybullet_robot_envs.envs.utils import goal_distance


class pandaReachGymEnv(PandaTaskBase):
    _TASK = _capi.TASK_REACH

    def __init__(self,
                 numControlledJoints=7,
                 use_IK=0,
                 action_repeat=1,
                 obj_name=get_objects_list()[1],
                 renders=False,
                 max_steps=1000,
                 obj_pose_rnd_std=0,
                 includeVelObs=True,
                 num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False, _lib=None, devices=None):
        device_id = devices if devices is not None else device_id
        self._setup(numControlledJoints, use_IK, action_repeat, obj_name, renders, max_steps, obj_pose_rnd_std,
                    0.0, includeVelObs, 0.03, num_envs, device_id, env_id_base, seed, _lib, auto_reset)

    def _distance(self):
        st = self._engine.get_state().astype(np.float64)
        ee = self._engine.observe()[:, :3].astype(np.float64)
        return goal_distance(ee, st[:, 9:12]), st

    def _termination(self):
        d, st = self._distance()
        done = (d <= self._target_dist_min) | (st[:, 36] != 0) | (st[:, 35] > self._max_steps)
        return self._squeeze(done.astype(np.float32))

    def _compute_reward(self):
        d, _ = self._distance()
        return self._squeeze(np.where(d <= self._target_dist_min, np.float32(1000.0) + (100 - d * 80), -d))
