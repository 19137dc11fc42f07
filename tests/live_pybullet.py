"""TEST INFRASTRUCTURE -- optional live cross-check against real PyBullet.

PyBullet is not installed in the build image nor (as far as observed) on the GPU boxes, so everything here is gated on
`import pybullet`.  Where it does import, this is the one mechanism that pins the physics restatement:

  * `model_to_urdf` writes the engine's own robot model (the committed parameter JSON, plus the stand-in collision spheres of
    model/table.py) as URDF text, so PyBullet simulates exactly the kinematic / inertial / collision model the engine steps;
  * `LivePandaPush` is a builder-written loop over the `pybullet` API that mirrors the reference's call sequence
    (R/envs/panda_envs/panda_push_gym_env.py:105-255: reset_simulation -> 100 + 100 + 1 settle steps, apply_action joint branch
    with POSITION_CONTROL kp 0.5 / kd 1.0, stepSimulation, observation, termination, reward), with time.sleep removed.

The product never imports this module."""
import math
import os
import tempfile

import numpy as np


def _rpy(R):
    """rotation matrix -> URDF fixed-axis roll, pitch, yaw (inverse of model/urdf.py: rpy_to_matrix)"""
    R = np.asarray(R, float)
    sp = -R[2, 0]
    if abs(sp) < 1 - 1e-12:
        return math.atan2(R[2, 1], R[2, 2]), math.asin(sp), math.atan2(R[1, 0], R[0, 0])
    return 0.0, math.copysign(math.pi / 2, sp), math.atan2(-R[0, 1], R[1, 1])


def model_to_urdf(model, spheres=()):
    """URDF text of a model dict (model/urdf.py format): one link per record, inertial frames at the COM in link axes, stand-in
    collision spheres (link name, centre, radius)."""
    def f(v):
        return " ".join("%.17g" % x for x in v)

    def link_xml(name, L):
        s = ['  <link name="%s">' % name]
        if L["mass"] > 0 or np.abs(np.asarray(L["inertia"])).max() > 0:
            I = np.asarray(L["inertia"], float)
            s.append('    <inertial><origin xyz="%s" rpy="0 0 0"/><mass value="%.17g"/>' % (f(L["com"]), L["mass"]))
            s.append('      <inertia ixx="%.17g" ixy="%.17g" ixz="%.17g" iyy="%.17g" iyz="%.17g" izz="%.17g"/></inertial>'
                     % (I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        for ln, c, r in [sp[:3] for sp in spheres]:
            if ln == name:
                s.append('    <collision><origin xyz="%s" rpy="0 0 0"/><geometry><sphere radius="%.17g"/></geometry></collision>' % (f(c), r))
        if L.get("lateral_friction") is not None:
            s.append('    <contact><lateral_friction value="%.17g"/></contact>' % L["lateral_friction"])
        s.append("  </link>")
        return "\n".join(s)

    out = ['<?xml version="1.0"?>', '<robot name="%s">' % model["name"], link_xml(model["base"]["name"], model["base"])]
    names = [model["base"]["name"]] + [l["name"] for l in model["links"]]
    jt = {0: "fixed", 1: "revolute", 2: "prismatic"}
    for i, L in enumerate(model["links"]):
        out.append(link_xml(L["name"], L))
        out.append('  <joint name="%s" type="%s"><parent link="%s"/><child link="%s"/>' % (L["joint_name"], jt[L["jtype"]], names[L["parent"] + 1], L["name"]))
        out.append('    <origin xyz="%s" rpy="%s"/><axis xyz="%s"/>' % (f(L["origin_xyz"]), f(_rpy(L["origin_R"])), f(L["axis"])))
        if L["jtype"]:
            out.append('    <limit lower="%.17g" upper="%.17g" effort="%.17g" velocity="%.17g"/><dynamics damping="%.17g"/>'
                       % (L["lower"], L["upper"], L["effort"], L["velocity"], L["damping"]))
        out.append("  </joint>")
    out.append("</robot>")
    return "\n".join(out) + "\n"


class LivePandaPush(object):
    """pandaPushGymEnv's simulation on real PyBullet (joint control, one env, DIRECT mode)."""
    HOME = {"panda_joint1": 0.0, "panda_joint2": -0.54, "panda_joint3": 0.0, "panda_joint4": -2.6, "panda_joint5": -0.30,
            "panda_joint6": 2.0, "panda_joint7": 1.0, "panda_finger_joint1": 0.02, "panda_finger_joint2": 0.02}

    def __init__(self, seed=0, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, max_steps=1000):
        import pybullet as p
        import pybullet_data
        from pybullet_robot_envs.model.table import panda_table, PANDA_SPHERES
        self.p, self.data = p, pybullet_data.getDataPath()
        self.cid = p.connect(p.DIRECT)
        _, model = panda_table()
        fd, self.urdf = tempfile.mkstemp(suffix=".urdf")
        os.write(fd, model_to_urdf(model, PANDA_SPHERES).encode())
        os.close(fd)
        self.rng = np.random.RandomState(seed)
        self.obj_std, self.tg_std, self.max_steps = obj_pose_rnd_std, tg_pose_rnd_std, max_steps
        self.ee_link = 11

    def reset(self):
        p, c = self.p, self.cid
        p.resetSimulation(physicsClientId=c)
        p.setPhysicsEngineParameter(numSolverIterations=150, physicsClientId=c)
        p.setTimeStep(1.0 / 240.0, physicsClientId=c)
        p.setGravity(0, 0, -9.8, physicsClientId=c)
        self.robot = p.loadURDF(self.urdf, basePosition=[0.0, 0.0, 0.625], useFixedBase=True,
                                flags=p.URDF_USE_INERTIA_FROM_FILE, physicsClientId=c)
        self.joints = []
        for i in range(p.getNumJoints(self.robot, physicsClientId=c)):
            info = p.getJointInfo(self.robot, i, physicsClientId=c)
            if info[2] in (p.JOINT_REVOLUTE, p.JOINT_PRISMATIC):
                q0 = self.HOME[info[1].decode()]
                self.joints.append((i, info[8], info[9]))
                p.resetJointState(self.robot, i, q0, physicsClientId=c)
                p.setJointMotorControl2(self.robot, i, p.POSITION_CONTROL, targetPosition=q0, positionGain=0.2, velocityGain=1.0, physicsClientId=c)
        for _ in range(100):
            p.stepSimulation(physicsClientId=c)
        p.loadURDF(os.path.join(self.data, "plane.urdf"), [0, 0, 0], physicsClientId=c)
        p.loadURDF(os.path.join(self.data, "table/table.urdf"), basePosition=[0.85, 0.0, 0.0], useFixedBase=True, physicsClientId=c)
        x, y, yaw = 0.45, 0.0, math.pi / 4          # WorldEnv._sample_pose (world_env.py:145-176)
        if self.obj_std > 0:
            x += self.rng.uniform(-self.obj_std, self.obj_std); y += self.rng.uniform(-self.obj_std, self.obj_std)
            yaw = self.rng.uniform(-math.pi / 4, math.pi / 4)
        self.obj = p.loadURDF(os.path.join(self.data, "cube_small.urdf"), basePosition=[x, y, 0.625 + 0.07],
                              baseOrientation=p.getQuaternionFromEuler([0, 0, yaw]), physicsClientId=c)
        for _ in range(101):
            p.stepSimulation(physicsClientId=c)
        op = p.getBasePositionAndOrientation(self.obj, physicsClientId=c)[0]
        self.target = np.array([op[0] + 0.05, op[1] + 0.05, op[2]])
        if self.tg_std > 0:
            self.target[:2] = np.asarray(op[:2]) + self.rng.normal(0, self.tg_std, 2)
            self.target[0] = min(max(self.target[0], 0.3 + 0.07), 0.65 - 0.07)
            self.target[1] = min(max(self.target[1], -0.3), 0.3)
        self.counter, self.terminated = 0, 0
        return self.state()

    def state(self):
        """(q[9], qd[9], object pos[3] + quat[4], object twist[6])"""
        p, c = self.p, self.cid
        js = p.getJointStates(self.robot, [j[0] for j in self.joints], physicsClientId=c)
        op, oq = p.getBasePositionAndOrientation(self.obj, physicsClientId=c)
        ov, ow = p.getBaseVelocity(self.obj, physicsClientId=c)
        return (np.array([s[0] for s in js]), np.array([s[1] for s in js]), np.array(list(op) + list(oq)), np.array(list(ov) + list(ow)))

    def step(self, action):
        p, c = self.p, self.cid
        q = self.state()[0]
        for k, (ji, lo, hi) in enumerate(self.joints[:7]):
            tgt = min(max(q[k] + 0.05 * float(action[k]), lo), hi)
            p.setJointMotorControl2(self.robot, ji, p.POSITION_CONTROL, targetPosition=tgt, positionGain=0.5, velocityGain=1.0, physicsClientId=c)
        p.stepSimulation(physicsClientId=c)
        ee = np.array(p.getLinkState(self.robot, self.ee_link, physicsClientId=c)[0])
        op = np.array(p.getBasePositionAndOrientation(self.obj, physicsClientId=c)[0])
        d1, d2 = np.linalg.norm(ee - op), np.linalg.norm(op - self.target)
        if d2 <= 0.1:
            self.terminated = 1
        done = bool(self.terminated or self.counter > self.max_steps)
        if not done:
            self.counter += 1
        reward = 1000.0 + (100.0 - 80.0 * d2) if d2 <= 0.1 else -d1 - d2
        return self.state(), reward, done

    def close(self):
        self.p.disconnect(self.cid)
        os.remove(self.urdf)
