#!/usr/bin/env python
"""Regenerate the committed model-parameter files from the reference's robot
description text (run in the dev container only; /root/reference is not on the
GPU box).  Output is DATA (parsed kinematic/inertial parameters as JSON), not a
copy of the URDF text.

    python tools/compile_models.py [/root/reference]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
from pybullet_robot_envs.model import urdf, table  # noqa: E402

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = os.path.join(ref, "pybullet_robot_envs/robot_data/franka_panda/panda_model.urdf")
m = urdf.parse_urdf(src, base_position=(0.0, 0.0, 0.625))
dst = os.path.join(ROOT, "pybullet-robot-envs_amd/pybullet_robot_envs/robot_data/franka_panda/panda_model.json")
table.save_model_json(m, dst)
print("wrote", dst, "links:", [l["name"] for l in m["links"]])

from pybullet_robot_envs.model import sdf  # noqa: E402
src = os.path.join(ref, "pybullet_robot_envs/robot_data/iCub/icub_model.sdf")
m = sdf.parse_sdf(src)
dst = os.path.join(ROOT, "pybullet-robot-envs_amd/pybullet_robot_envs/robot_data/iCub/icub_model.json")
table.save_model_json(m, dst)
print("wrote", dst, "links:", len(m["links"]), "dof:", sum(1 for l in m["links"] if l["jtype"]))

src = os.path.join(ref, "pybullet_robot_envs/robot_data/iCub/icub_model_with_hands.sdf")
m = sdf.parse_sdf(src)
dst = os.path.join(ROOT, "pybullet-robot-envs_amd/pybullet_robot_envs/robot_data/iCub/icub_model_with_hands.json")
table.save_model_json(m, dst)
print("wrote", dst, "links:", len(m["links"]), "dof:", sum(1 for l in m["links"] if l["jtype"]))
