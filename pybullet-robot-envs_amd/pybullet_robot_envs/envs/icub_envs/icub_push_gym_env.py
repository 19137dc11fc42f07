"""iCubPushGymEnv (reference pybullet_robot_envs/envs/icub_envs/icub_push_gym_env.py): push the object to the target;
success when the object is within 0.03 m of the target; reward_type 0: -d1 - d2 (+1000), 1: normalised (+1000)."""
import numpy as np

from pybullet_robot_envs import _capi
from pybullet_robot_envs.envs.icub_envs._base import ICubTaskBase
from pybullet_robot_envs.envs.world_envs.world_env import get_objects_list
from pybullet_robot_envs.envs.utils import goal_distance


class iCubPushGymEnv(ICubTaskBase):
    _TASK = _capi.TASK_PUSH

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
                 num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False, _lib=None, devices=None, floating_base=False):
        device_id = devices if devices is not None else device_id
        self._setup_icub(action_repeat, use_IK, control_arm, control_orientation, obj_name, obj_pose_rnd_std, tg_pose_rnd_std,
                         renders, max_steps, reward_type, num_envs, device_id, env_id_base, seed, _lib, auto_reset, floating_base)

    @property
    def _init_dist_hand_obj(self):
        return self._squeeze(self._engine.get_state()[:, self._engine.x_off + 12].astype(np.float64))

    @property
    def _max_dist_obj_tg(self):
        return self._squeeze(self._engine.get_state()[:, self._engine.x_off + 13].astype(np.float64))

    def _distances(self):
        eng = self._engine
        st = eng.get_state().astype(np.float64)
        ee = eng.observe()[:, :3].astype(np.float64)
        ob = st[:, eng.obj_off:eng.obj_off + 3]
        return goal_distance(ee, ob), goal_distance(ob, st[:, eng.x_off:eng.x_off + 3]), st

    def _termination(self):
        d1, d2, st = self._distances()
        x = self._engine.x_off
        done = (d2 <= self._target_dist_min) | (st[:, x + 4] != 0) | (st[:, x + 3] > self._max_steps)
        return self._squeeze(done.astype(np.float32))

    def _compute_reward(self):
        d1, d2, st = self._distances()
        x = self._engine.x_off
        if self._reward_type == 0:
            reward = -d1 - d2
        else:
            rew1, rew2 = 0.125, 0.25
            reward = rew1 * (1 - d1 / st[:, x + 12]) + np.where(d1 > 0.1, 0.0, rew2 * (1 - d2 / st[:, x + 13]))
        return self._squeeze(reward + np.where(d2 <= self._target_dist_min, np.float32(1000.0), 0.0))
