"""iCubHandsEnv -- the iCub with articulated hands (reference pybullet_robot_envs/envs/icub_envs/icub_env_with_hands.py).

Same constructor signature, attributes and methods as the reference class, for a whole batch of environments: every method
that talks to PyBullet there (`apply_action`, `open_hand`, `pre_grasp`, `grasp`, `check_contact_fingertips`,
`check_collision`, `get_observation`) talks to the GPU engine here (csrc/pbre_hands.hip: one env per wavefront, 60
simulated DoF -- the 72-DoF model without its legs, which a fixed base decouples exactly).  The reference has no task env
for this robot, only the scripted demo examples/helloworlds/helloworld_icub.py; its scene (table at x = 1, a brick-sized
object dropped at (0.5, -0.03)) is the engine's default world for this robot, and `step_simulation(n)` stands for the demo's
`for _ in range(n): p.stepSimulation()` loops.  PyBullet's motors keep their last command, so do the engine's (a per-env
motor record on the GPU): `apply_action` / the finger commands only write commands, `step_simulation` advances time, and
`step(action)` is the fused command + one step + observation used for throughput.

Hand-pose commands may be 3 values (position), 6 (position + Euler angles, clipped to the arm's Euler limits as in the
reference) or 7 (position + quaternion, used as given, as in the reference).  `max_vel` is the motors' `maxVelocity` (the rhs
clamp of Bullet's motor row [EXT-UNVERIFIED]); control_eu_or_quat=1 returns quaternion observations (converted on the host)."""
import math as m

import numpy as np

from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs.envs.icub_envs.icub_env import iCubEnv
from pybullet_robot_envs.model.table import icub_hands_table, hand_joint_names, HAND_TIPS, GRASP_POS


class iCubHandsEnv(iCubEnv):

    initial_positions = {n: 0.0 for s in ("l", "r") for n in hand_joint_names(s)}
    initial_positions.update(iCubEnv.initial_positions)

    joint_groups = {'l_hand': hand_joint_names('l'), 'r_hand': hand_joint_names('r')}
    joint_groups.update(iCubEnv.joint_groups)

    def __init__(self, physicsClientId, use_IK=0, control_arm='l', control_orientation=1, control_eu_or_quat=0):

        self._physics_client_id = physicsClientId
        self._client = _client.get(physicsClientId)
        self._client.robot = self
        self._use_IK = use_IK
        self._control_orientation = control_orientation
        self._control_eu_or_quat = control_eu_or_quat

        self._home_hand_pose = []
        self._home_motor_pose = []

        self._grasp_pos = list(GRASP_POS)

        self._workspace_lim = [[0.15, 0.50], [-0.3, 0.3], [0.5, 1.0]]
        self._eu_lim = [[-m.pi, m.pi], [-m.pi, m.pi], [-m.pi, m.pi]]

        self._control_arm = control_arm if control_arm == 'r' or control_arm == 'l' else 'l'  # left arm by default
        self._joints_to_control = []
        self._joints_to_block = []
        self._joint_name_to_ids = {}

        self.robot_id = 0
        self.end_eff_idx = []

        # set initial hand pose (icub_env_with_hands.py:76-81)
        if self._control_arm == 'l':
            self._home_hand_pose = [0.2, 0.3, 0.8, -m.pi, 0, -m.pi/2]   # x,y,z, roll,pitch,yaw
            self._eu_lim = [[-3/2*m.pi, -m.pi/2], [-m.pi / 2, m.pi / 2], [0, -m.pi]]
        else:
            self._home_hand_pose = [0.2, -0.3, 0.8, 0, 0,  m.pi/2]
            self._eu_lim = [[-m.pi / 2, m.pi / 2], [-m.pi / 2, m.pi / 2], [0, m.pi]]

        self._last_out = None
        self.seed()
        self.reset()

    # ------------------------------------------------------------------ model bookkeeping + engine
    def reset(self):
        # replaces p.loadSDF("icub_model_with_hands.sdf") + p.createConstraint (icub_env_with_hands.py:87-102): parsed
        # parameters -> RobotTable; the reference's joint names / indices cover the whole model (78 joints), the engine
        # simulates it without the legs (60 DoF)
        self.robot_table, self._sim_model, self._info = icub_hands_table(self._control_arm)
        _, self._model, self._full_info = icub_hands_table(self._control_arm, full=True)
        self._num_joints = len(self._model["links"])
        self._joint_name_to_ids = {}
        for i, link in enumerate(self._model["links"]):
            if link["jtype"] != 0:
                assert link["joint_name"] in self.initial_positions.keys()
                self._joint_name_to_ids[link["joint_name"]] = i

        # Controlled joints (reference :123-146; its conditions read `a or b and c`, i.e. `in l_arm or (in l_hand and arm == 'l')`):
        # torso and BOTH arms always, the hand of control_arm; link-index order.  End effector: that arm's wrist-yaw link.
        if not self._joints_to_control:
            side = 'l' if self._control_arm == 'l' else 'r'
            driven = set(self.joint_groups['torso']) | set(self.joint_groups['l_arm']) | set(self.joint_groups['r_arm']) \
                | set(self.joint_groups[side + '_hand'])
            ids = self._joint_name_to_ids
            self._joints_to_control = [i for name, i in ids.items() if name in driven]
            self._joints_to_block = [i for name, i in ids.items() if name not in driven]
            self.end_eff_idx = ids[side + '_wrist_yaw']
        assert self.end_eff_idx == self._full_info["ee_link"]
        assert self.controlled_dofs() == self._info["controlled"]

        self.ll, self.ul, self.jr, self.rs, self.jd = self.get_joint_ranges()

        # motors at the initial positions (gain 0.2), `apply_action(home_hand_pose)` + one step when use_IK (:108-121, 154-156),
        # then the demo's settle with the world loaded: all inside pbre_reset
        self._build_engine()
        self._last_out = None
        self._engine.reset()

    def _build_engine(self):
        c = self._client
        if c.engine is not None:
            c.engine.close()
        dofs = self.controlled_dofs()
        overrides = dict(device_id=c.device_id, env_id_base=c.env_id_base, seed=c.seed,
                         use_ik=1 if self._use_IK else 0, control_orientation=1 if self._control_orientation else 0,
                         num_controlled_joints=len(dofs), num_joints_ctrl=len(dofs), act_dof=dofs + [-1] * (64 - len(dofs)),
                         home=self.sim_home(), home_hand_pose=[float(x) for x in self._home_hand_pose],
                         eu_lim=[-1e9, 1e9] * 3,        # Euler limits are applied here (apply_action): a quaternion command bypasses them
                         ik_link_offset=list(self._com_to_link_hand_frame()[0]),
                         robot_ws=[x for lim in self._workspace_lim for x in lim])
        c.engine = _capi.make_engine(self.robot_table, devices=c.devices, task=_capi.TASK_REACH, num_envs=c.num_envs, lib=c.lib,
                                robot=_capi.ROBOT_ICUB_HANDS, **overrides)
        self._engine = c.engine
        assert self._engine.act_dim == (len(dofs) if not self._use_IK else (6 if self._control_orientation else 3)) and self._engine.state_floats == 272

    def _com_to_link_hand_frame(self):
        if self._control_arm == 'r':
            com_T_link_hand = ((-0.011682, 0.051682, -0.000577), (0.0, 0.0, 0.0, 1.0))
        else:
            com_T_link_hand = ((-0.011682, 0.051355, 0.000577), (0.0, 0.0, 0.0, 1.0))
        return com_T_link_hand

    # ------------------------------------------------------------------ commands
    # apply_action / step_simulation / the 3-6-7 hand-pose command forms are iCubEnv's (icub_env.py), on this class's own engine
    def _engine_or_build(self):
        return self._engine

    def _robot_level(self):
        return self._engine

    def step(self, action):
        """Fused apply_action + one simulation step + observation for the whole batch (one kernel launch)."""
        a = self._batch(action)
        if self._use_IK:
            a = self._hand_pose_command(a, self._engine)
        obs, rew, done = self._engine.step(a)
        self._last_out = obs
        return obs

    def _finger_dofs(self):
        return self._info["fingers"]

    def open_hand(self, env_mask=None):
        # open fingers (icub_env_with_hands.py:167-182)
        self._engine.set_motors(self._finger_dofs(), [0.0] * 20, 0.1, 0.0, env_mask)

    def pre_grasp(self, env_mask=None):
        # move fingers to pre-grasp configuration: thumb opposition at 1.57 (:184-204)
        names = self.joint_groups[self._control_arm + '_hand']
        thumb = self._control_arm + '_hand::' + self._control_arm + '_tj2'
        pos = [1.57 if n == thumb else 0.0 for n in names]
        self._engine.set_motors(self._finger_dofs(), pos, 0.1, 0.0, env_mask)

    def grasp(self, pos=None, env_mask=None):
        # close fingers: position control towards `pos`, gain 0.1, force 10 (:206-244)
        if pos is None:
            pos = self._grasp_pos
        if len(pos) != 20:
            raise AssertionError('grasp needs one target per finger joint (20)', len(pos))
        self._engine.set_motors(self._finger_dofs(), [float(x) for x in pos], 0.1, 10.0, env_mask)

    # ------------------------------------------------------------------ queries
    def get_action_dim(self):
        if not self._use_IK:
            return len(self._joints_to_control)
        if self._control_orientation:
            return 6 if self._control_eu_or_quat == 0 else 7      # the engine takes Euler angles; a quaternion command is converted
        return 3

    def _tail(self):
        return self._engine.observe()[:, -7:].astype(np.float64)

    def check_contact_fingertips(self, obj_id=None):
        """Number of fingertips of the controlled hand touching the object and the mean normal force on each of the five
        (index, little, middle, ring, thumb: the 4th joint of every finger, tips_idxs :248).  ints / lists for one env, arrays
        [N] / [N, 5] for a batch."""
        t = self._tail()
        n, f = t[:, 5].astype(int), t[:, :5]
        if self.num_envs == 1:
            return int(n[0]), list(f[0])
        return n, f

    def check_collision(self, obj_id=None):
        # any contact with the object that is not a fingertip contact (:310-318)
        t = self._tail()
        c = (t[:, 6] - t[:, 5]) > 0
        return bool(c[0]) if self.num_envs == 1 else c

    def fingertip_indices(self):
        names = self.joint_groups[self._control_arm + '_hand']
        return [self._joint_name_to_ids[names[k]] for k in HAND_TIPS]
