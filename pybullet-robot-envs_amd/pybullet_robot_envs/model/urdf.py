"""URDF -> generic kinematic-tree description (host side, runs once).

The reference hands `panda_model.urdf` to `p.loadURDF(..., useFixedBase=True,
flags=URDF_USE_INERTIA_FROM_FILE | ...)` (reference
pybullet_robot_envs/envs/panda_envs/panda_env.py:53-56) and PyBullet builds a
multibody from it.  This module is the engine's own parser: it reads the URDF
text and produces a plain dict ("model") that `flatten.py` turns into the flat
RobotTable consumed by the C-ABI (include/pbre.h) and by the test oracle.

Link order follows PyBullet's convention of depth-first traversal in file
order of the joints so that link index i == joint index i (SURVEY Appendix A:
`end_eff_idx = 11` must be `panda_grasptarget`).
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1, 2
_JT = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE, "continuous": JOINT_REVOLUTE,
       "prismatic": JOINT_PRISMATIC}


def rpy_to_matrix(rpy):
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix (child->parent)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _vec(s, n=3, default=None):
    if s is None:
        return np.array(default if default is not None else [0.0] * n, dtype=float)
    return np.array([float(x) for x in s.split()], dtype=float)


def _origin(el):
    o = el.find("origin") if el is not None else None
    if o is None:
        return np.zeros(3), np.zeros(3)
    return _vec(o.get("xyz")), _vec(o.get("rpy"))


def _parse_link(el):
    out = {"name": el.get("name"), "mass": 0.0, "com": [0.0, 0.0, 0.0],
           "inertia": np.zeros((3, 3)).tolist(), "lateral_friction": None,
           "has_collision": el.find("collision") is not None}
    ine = el.find("inertial")
    if ine is not None:
        xyz, rpy = _origin(ine)
        m = ine.find("mass")
        out["mass"] = float(m.get("value")) if m is not None else 0.0
        out["com"] = xyz.tolist()
        i = ine.find("inertia")
        if i is not None:
            g = lambda k: float(i.get(k, 0.0))
            I = np.array([[g("ixx"), g("ixy"), g("ixz")],
                          [g("ixy"), g("iyy"), g("iyz")],
                          [g("ixz"), g("iyz"), g("izz")]])
            R = rpy_to_matrix(rpy)
            out["inertia"] = (R @ I @ R.T).tolist()  # expressed in link axes, about the COM
    c = el.find("contact")
    if c is not None:
        lf = c.find("lateral_friction")
        if lf is not None:
            out["lateral_friction"] = float(lf.get("value"))
    return out


def parse_urdf(path, base_position=(0.0, 0.0, 0.0)):
    root = ET.parse(path).getroot()
    links = {l.get("name"): _parse_link(l) for l in root.findall("link")}
    joints = []
    children = set()
    for j in root.findall("joint"):
        xyz, rpy = _origin(j)
        ax = j.find("axis")
        lim = j.find("limit")
        dyn = j.find("dynamics")
        joints.append({
            "name": j.get("name"), "type": _JT[j.get("type")],
            "parent": j.find("parent").get("link"), "child": j.find("child").get("link"),
            "xyz": xyz.tolist(), "rpy": rpy.tolist(),
            "axis": (_vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0, 0])).tolist(),
            "lower": float(lim.get("lower", 0.0)) if lim is not None else 0.0,
            "upper": float(lim.get("upper", 0.0)) if lim is not None else 0.0,
            "effort": float(lim.get("effort", 0.0)) if lim is not None else 0.0,
            "velocity": float(lim.get("velocity", 0.0)) if lim is not None else 0.0,
            "damping": float(dyn.get("damping", 0.0)) if dyn is not None else 0.0,
        })
        children.add(j.find("child").get("link"))
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, "URDF must have exactly one root link"
    base = roots[0]

    # depth-first in file order (PyBullet's URDF2Bullet visits child joints in file order)
    order = []

    def visit(link_name, parent_idx):
        for j in joints:
            if j["parent"] == link_name:
                idx = len(order)
                order.append((j, parent_idx))
                visit(j["child"], idx)

    visit(base, -1)
    out_links = []
    for j, parent_idx in order:
        L = dict(links[j["child"]])
        ax = np.array(j["axis"], dtype=float)
        n = np.linalg.norm(ax)
        if n > 0:
            ax = ax / n
        L.update({"joint_name": j["name"], "jtype": j["type"], "parent": parent_idx,
                  "axis": ax.tolist(), "origin_xyz": j["xyz"],
                  "origin_R": rpy_to_matrix(j["rpy"]).tolist(),
                  "lower": j["lower"], "upper": j["upper"], "effort": j["effort"],
                  "velocity": j["velocity"], "damping": j["damping"]})
        out_links.append(L)
    return {"name": root.get("name"), "base": dict(links[base]),
            "base_position": list(map(float, base_position)),
            "base_R": np.eye(3).tolist(), "fixed_base": True, "links": out_links}
