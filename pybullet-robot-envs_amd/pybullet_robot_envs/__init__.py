"""pybullet_robot_envs -- MI355X-native drop-in for the env.step() hot path of hsp-iit/pybullet-robot-envs.

Same module layout, class names, constructor signatures and gym ids as the reference package
(reference pybullet_robot_envs/__init__.py:7-80); PyBullet is replaced by the batched HIP engine
libpbre.so (csrc/, C-ABI in include/pbre.h).  `renders` is accepted but there is no GUI."""
from pybullet_robot_envs._gym import register
from pybullet_robot_envs._client import connect, disconnect  # noqa: F401

_IDS = [
    ('iCubReach-v0', 'iCubReachGymEnv', {'use_IK': 1, 'control_arm': 'l', 'control_orientation': 0,
                                         'obj_pose_rnd_std': 0, 'max_steps': 1000, 'renders': True}),
    ('iCubPush-v0', 'iCubPushGymEnv', {'use_IK': 1, 'control_arm': 'l', 'control_orientation': 0,
                                       'obj_pose_rnd_std': 0.05, 'tg_pose_rnd_std': 0, 'max_steps': 1000,
                                       'reward_type': 0, 'renders': True}),
    ('iCubPushGoal-v0', 'iCubPushGymGoalEnv', {'use_IK': 1, 'control_arm': 'r', 'control_orientation': 1,
                                               'obj_pose_rnd_std': 0.05, 'tg_pose_rnd_std': 0, 'max_steps': 1000,
                                               'renders': True}),
    ('pandaReach-v0', 'pandaReachGymEnv', {'use_IK': 0, 'obj_pose_rnd_std': 0.05, 'max_steps': 1000,
                                           'includeVelObs': True, 'renders': True}),
    ('pandaPush-v0', 'pandaPushGymEnv', {'use_IK': 0, 'obj_pose_rnd_std': 0.05, 'tg_pose_rnd_std': 0,
                                         'includeVelObs': True, 'max_steps': 1000, 'renders': True}),
    ('pandaPushGoal-v0', 'pandaPushGymGoalEnv', {'use_IK': 0, 'obj_pose_rnd_std': 0.05, 'tg_pose_rnd_std': 0,
                                                 'includeVelObs': True, 'max_steps': 1000, 'renders': True}),
]
for _id, _cls, _kw in _IDS:
    try:
        register(id=_id, entry_point='pybullet_robot_envs.envs:' + _cls, max_episode_steps=1000, kwargs=_kw)
    except Exception:  # already registered (module re-import under real gym)
        pass
