"""iCubEnv / iCubHandsEnv placeholders (reference icub_env.py, icub_env_with_hands.py); see icub_reach_gym_env.py."""


class iCubEnv(object):
    def __init__(self, physicsClientId, use_IK=0, control_arm='l', control_orientation=1, control_eu_or_quat=0):
        raise NotImplementedError("iCubEnv is not implemented by the MI355X engine yet (DESIGN.md 'Out of scope')")
