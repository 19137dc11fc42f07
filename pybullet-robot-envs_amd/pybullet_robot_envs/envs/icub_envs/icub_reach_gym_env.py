"""iCubReachGymEnv (reference pybullet_robot_envs/envs/icub_envs/icub_reach_gym_env.py): reach the object with the hand;
success when the hand COM is within 0.03 m of the object; reward -d, plus 1000 + (100 - 80 d) on success."""
import numpy as np

from pybullet_robot_envs import _capi
from pybullet_robot_envs.envs.icub_envs._base import ICubTaskBase
from pybullet_robot_envs.envs.world_envs.world_env import get_objects_list
from pybullet_robot_envs.envs.utils import goal_distance


class iCubReachGymEnv(ICubTaskBase):
    _TASK = _capi.TASK_REACH

    def __init__(self,
                 action_repeat=1,
                 use_IK=1,
                 control_arm='l',
                 control_orientation=0,
                 obj_name=get_objects_list()[0],
                 obj_pose_rnd_std=0,
                 renders=False,
                 max_steps=2000,
                 num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False, _lib=None, devices=None, floating_base=False):
        device_id = devices if devices is not None else device_id
        self._setup_icub(action_repeat, use_IK, control_arm, control_orientation, obj_name, obj_pose_rnd_std, 0.0,
                         renders, max_steps, 1, num_envs, device_id, env_id_base, seed, _lib, auto_reset, floating_base)

    def _distance(self):
        eng = self._engine
        st = eng.get_state().astype(np.float64)
        ee = eng.observe()[:, :3].astype(np.float64)
        return goal_distance(ee, st[:, eng.obj_off:eng.obj_off + 3]), st

    def _termination(self):
        d, st = self._distance()
        x = self._engine.x_off
        done = (d <= self._target_dist_min) | (st[:, x + 4] != 0) | (st[:, x + 3] > self._max_steps)
        return self._squeeze(done.astype(np.float32))

    def _compute_reward(self):
        d, _ = self._distance()
        return self._squeeze(-d + np.where(d <= self._target_dist_min, np.float32(1000.0) + (100 - d * 80), 0.0))
