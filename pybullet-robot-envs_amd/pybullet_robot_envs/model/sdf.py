"""SDF -> generic kinematic-tree description (host side, runs once).

The reference hands `icub_model.sdf` to `p.loadSDF` (reference
pybullet_robot_envs/envs/icub_envs/icub_env.py:91-92): PyBullet builds a floating-base
multibody whose base is the link without a parent joint, then the reference pins that base
to the world with a fixed constraint (icub_env.py:97-103).  This parser reads the SDF text
and produces the same plain dict ("model") as `urdf.parse_urdf`.

SDF conventions used (SDF 1.7, no `use_parent_model_frame` in the file): a link <pose> is
expressed in the model frame, the model <pose> in the world frame; the joint frame is the
child link frame (all joint <pose> are zero) and <axis><xyz> is expressed in it; an
<inertial><pose> places the COM frame in the link frame.

Link order: depth-first from the base, child joints in file order (link index i == joint
index i, as PyBullet numbers them [EXT-UNVERIFIED]).  The reference only addresses joints
by name (icub_env.py:107-150), so the order matters solely for the order of the controlled
joints in the action / observation vectors, which is this traversal order.
"""
import xml.etree.ElementTree as ET

import numpy as np

from pybullet_robot_envs.model.urdf import rpy_to_matrix, JOINT_FIXED, JOINT_REVOLUTE, JOINT_PRISMATIC

_JT = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE, "prismatic": JOINT_PRISMATIC}


def _pose(el):
    """<pose>x y z r p y</pose> -> (R, p); identity when absent."""
    if el is None or el.text is None:
        return np.eye(3), np.zeros(3)
    v = [float(x) for x in el.text.split()]
    return rpy_to_matrix(v[3:6]), np.array(v[:3])


def _f(el, tag, default=0.0):
    e = el.find(tag) if el is not None else None
    return float(e.text) if e is not None and e.text is not None else default


def parse_sdf(path):
    model = ET.parse(path).getroot().find("world").find("model")
    Rm, pm = _pose(model.find("pose"))
    links = {}
    for l in model.findall("link"):
        Rl, pl = _pose(l.find("pose"))
        ine = l.find("inertial")
        Ri, pi = _pose(ine.find("pose") if ine is not None else None)
        I = np.zeros((3, 3))
        if ine is not None and ine.find("inertia") is not None:
            i = ine.find("inertia")
            g = lambda k: _f(i, k)
            I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        links[l.get("name")] = {"name": l.get("name"), "R": Rl, "p": pl, "mass": _f(ine, "mass"), "com": pi.tolist(),
                                "inertia": (Ri @ I @ Ri.T).tolist(), "lateral_friction": None,
                                "has_collision": l.find("collision") is not None}
    joints, children = [], set()
    for j in model.findall("joint"):
        ax = j.find("axis")
        lim = ax.find("limit") if ax is not None else None
        dyn = ax.find("dynamics") if ax is not None else None
        jt = _JT[j.get("type")]
        joints.append({"name": j.get("name"), "type": jt, "parent": j.find("parent").text, "child": j.find("child").text,
                       "axis": [float(x) for x in ax.find("xyz").text.split()] if ax is not None else [1.0, 0.0, 0.0],
                       "lower": _f(lim, "lower") if jt != JOINT_FIXED else 0.0, "upper": _f(lim, "upper") if jt != JOINT_FIXED else 0.0,
                       "effort": _f(lim, "effort"), "velocity": _f(lim, "velocity"),
                       "damping": _f(dyn, "damping") if jt != JOINT_FIXED else 0.0})
        children.add(j.find("child").text)
    roots = [n for n in links if n not in children]
    assert len(roots) == 1, "SDF model must have exactly one root link"
    base = roots[0]
    order = []

    def visit(name, parent_idx):
        for j in joints:
            if j["parent"] == name:
                idx = len(order)
                order.append((j, parent_idx))
                visit(j["child"], idx)

    visit(base, -1)
    out = []
    for j, parent_idx in order:
        c, p = links[j["child"]], links[j["parent"]]
        R = p["R"].T @ c["R"]                      # child link frame in the parent link frame (zero configuration)
        t = p["R"].T @ (c["p"] - p["p"])
        ax = np.array(j["axis"], dtype=float)
        n = np.linalg.norm(ax)
        ax = ax / n if n > 0 else ax
        L = {k: c[k] for k in ("name", "mass", "com", "inertia", "lateral_friction", "has_collision")}
        L.update({"joint_name": j["name"], "jtype": j["type"], "parent": parent_idx, "axis": ax.tolist(),
                  "origin_xyz": t.tolist(), "origin_R": R.tolist(), "lower": j["lower"], "upper": j["upper"],
                  "effort": j["effort"], "velocity": j["velocity"], "damping": j["damping"]})
        out.append(L)
    b = links[base]
    Rb, pb = Rm @ b["R"], pm + Rm @ b["p"]         # base link frame in the world as loaded
    base_d = {k: b[k] for k in ("name", "mass", "com", "inertia", "lateral_friction", "has_collision")}
    return {"name": model.get("name"), "base": base_d, "base_position": pb.tolist(), "base_R": Rb.tolist(),
            "fixed_base": False, "links": out}
