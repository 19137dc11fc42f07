"""Host-side contact query on a downloaded batch state -- `WorldEnv.check_contact` (reference world_env.py:128-134:
`len(p.getContactPoints(obj_id, body_id)) > 0`).  The GPU step keeps its contact points in registers; this restates its
*detection* rule on the host for the rare caller that asks: a pair is "in contact" when its distance is below the engine's
contact margin (Bullet reports points inside the contact-breaking threshold the same way).  Geometry is the engine's own: the
robot's stand-in collision spheres of the RobotTable (model/table.py), the object box and the table box of pbre_physics.
numpy, vectorised over the batch; float64 (the device tests in float32, so a pair within rounding of the margin may differ)."""
import numpy as np

from pybullet_robot_envs.model.table import HEADER, LINK_STRIDE, SPHERE_STRIDE

OBJECT_TABLE, ROBOT_OBJECT, ROBOT_TABLE = 1, 2, 4


def _quat_R(q):
    x, y, z, w = (q[:, k] for k in range(4))
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _axis_R(axis, ang):
    """Rodrigues rotation about the unit vector `axis` by the angles ang[N] -> [N, 3, 3]"""
    a = np.asarray(axis, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    s, c = np.sin(ang)[:, None, None], np.cos(ang)[:, None, None]
    return np.eye(3)[None] + s * K[None] + (1 - c) * (K @ K)[None]


def link_frames(table, q):
    """World rotation [N, nl, 3, 3] and origin [N, nl, 3] of every link frame of the RobotTable for joint positions q[N, ndof]."""
    table = np.asarray(table, float)
    nl = int(table[2])
    n = q.shape[0]
    Rb, pb = table[9:18].reshape(3, 3), table[6:9]
    R = np.empty((n, nl, 3, 3)); p = np.empty((n, nl, 3))
    for i in range(nl):
        r = table[HEADER + i * LINK_STRIDE: HEADER + (i + 1) * LINK_STRIDE]
        par, jt, axis, xyz, R0, dof = int(r[0]), int(r[1]), r[2:5], r[5:8], r[8:17].reshape(3, 3), int(r[33])
        Rp = R[:, par] if par >= 0 else np.broadcast_to(Rb, (n, 3, 3))
        pp = p[:, par] if par >= 0 else np.broadcast_to(pb, (n, 3))
        if jt == 1:                       # revolute
            Rl = R0[None] @ _axis_R(axis, q[:, dof]); pl = np.broadcast_to(xyz, (n, 3))
        elif jt == 2:                     # prismatic
            Rl = np.broadcast_to(R0, (n, 3, 3)); pl = xyz[None] + (R0 @ axis)[None] * q[:, dof, None]
        else:
            Rl = np.broadcast_to(R0, (n, 3, 3)); pl = np.broadcast_to(xyz, (n, 3))
        R[:, i] = Rp @ Rl
        p[:, i] = pp + np.einsum("nij,nj->ni", Rp, pl)
    return R, p


def _sphere_box_dist(sc, sr, bc, Rb, h):
    """signed distance of spheres (centres sc[N, 3], radius sr) to the boxes (centre bc[N, 3], rotation Rb[N, 3, 3], half extents h)"""
    dl = np.einsum("nji,nj->ni", Rb, sc - bc)
    cl = np.clip(dl, -h, h)
    ln = np.linalg.norm(dl - cl, axis=1)
    best = (h - np.abs(dl)).min(axis=1)
    return np.where(ln < 1e-9, -best - sr, ln - sr)


def contact_flags(table, state, ndof, phys, no_object=False):
    """[N] uint8 of OBJECT_TABLE | ROBOT_OBJECT | ROBOT_TABLE for the batch state records state[N, F] (Q | V | X layout of
    include/pbre.h: joints at [0, ndof), object position / quaternion behind them); phys = pbre_physics (Engine.get_physics())."""
    table = np.asarray(table, float)
    st = np.asarray(state, float)
    n = st.shape[0]
    nl, ns = int(table[2]), int(table[5])
    R, p = link_frames(table, st[:, :ndof])
    margin = float(phys.contact_margin)
    tc, th = np.array(list(phys.table_c), float), np.array(list(phys.table_h), float)
    oh = np.array(list(phys.obj_h), float)
    op, Ro = st[:, ndof:ndof + 3], _quat_R(st[:, ndof + 3:ndof + 7])
    eye = np.broadcast_to(np.eye(3), (n, 3, 3))
    flags = np.zeros(n, np.uint8)
    base = HEADER + nl * LINK_STRIDE
    for k in range(ns):
        s = table[base + k * SPHERE_STRIDE: base + (k + 1) * SPHERE_STRIDE]
        li, c, rad = int(s[0]), s[1:4], float(s[4])
        sc = p[:, li] + np.einsum("nij,j->ni", R[:, li], c)
        if not no_object:
            flags |= np.where(_sphere_box_dist(sc, rad, op, Ro, oh) < margin, ROBOT_OBJECT, 0).astype(np.uint8)
        flags |= np.where(_sphere_box_dist(sc, rad, np.broadcast_to(tc, (n, 3)), eye, th) < margin, ROBOT_TABLE, 0).astype(np.uint8)
    if not no_object:                     # the box's vertices against the table top (pbre_fast.hpp: the object rows' candidates)
        top, bot = tc[2] + th[2], tc[2] - th[2]
        for v in range(8):
            l = np.array([oh[0] if v & 1 else -oh[0], oh[1] if v & 2 else -oh[1], oh[2] if v & 4 else -oh[2]])
            x = op + np.einsum("nij,j->ni", Ro, l)
            on = (np.abs(x[:, 0] - tc[0]) <= th[0]) & (np.abs(x[:, 1] - tc[1]) <= th[1]) & (x[:, 2] > bot)
            flags |= np.where(on & (x[:, 2] - top < margin), OBJECT_TABLE, 0).astype(np.uint8)
    return flags
