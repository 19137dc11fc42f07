"""iCubPushGymGoalEnv -- name kept importable for API parity (reference pybullet_robot_envs/envs/icub_envs/icub_push_gym_goal_env.py).

The iCub tasks (32-DoF floating-base humanoid held by a fixed constraint, IK control by default,
reference R/__init__.py:7-43) are not implemented by the batched HIP engine yet: its kernel maps one DoF
per lane of a 16-lane group (<= 9 robot DoF, fixed base).  SURVEY 8(f) / DESIGN.md list this as next."""


class iCubPushGymGoalEnv(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("iCubPushGymGoalEnv: the iCub environments are not implemented by the MI355X engine yet "
                                  "(Panda reach/push/push-goal are); see DESIGN.md, section 'Out of scope'")
