"""Task environments of the batched engine, exported under the names the reference package exports."""
from pybullet_robot_envs.envs.panda_envs.panda_push_gym_env import pandaPushGymEnv
from pybullet_robot_envs.envs.panda_envs.panda_push_gym_goal_env import pandaPushGymGoalEnv
from pybullet_robot_envs.envs.panda_envs.panda_reach_gym_env import pandaReachGymEnv
from pybullet_robot_envs.envs.icub_envs.icub_push_gym_env import iCubPushGymEnv
from pybullet_robot_envs.envs.icub_envs.icub_push_gym_goal_env import iCubPushGymGoalEnv
from pybullet_robot_envs.envs.icub_envs.icub_reach_gym_env import iCubReachGymEnv

__all__ = ["pandaReachGymEnv", "pandaPushGymEnv", "pandaPushGymGoalEnv", "iCubReachGymEnv", "iCubPushGymEnv", "iCubPushGymGoalEnv"]
