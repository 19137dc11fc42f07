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


def _sphere_round_dist(shape, sc, sr, bc, Rb, h):
    """signed distance of spheres to a round object (shape 1 sphere of radius h[0], 2 cylinder about local z: radius h[0], half height
    h[2]) -- csrc/pbre_objstep.hpp: Shapes::sphere_round"""
    d = sc - bc
    if shape == 1:
        return np.linalg.norm(d, axis=1) - h[0] - sr
    dl = np.einsum("nji,nj->ni", Rb, d)
    rho = np.hypot(dl[:, 0], dl[:, 1])
    rc, zc = np.minimum(rho, h[0]), np.clip(dl[:, 2], -h[2], h[2])
    ln = np.hypot(rho - rc, dl[:, 2] - zc)
    inside = -np.minimum(h[0] - rho, h[2] - np.abs(dl[:, 2]))
    return np.where(ln < 1e-9, inside - sr, ln - sr)


def _closest_on_triangles(p, A, B, C):
    """closest points of the triangles (A, B, C: [T, 3]) to the points p[N, 3] -> squared distances [N, T] (Ericson 5.1.5, vectorised:
    the regions are applied from the lowest priority -- the face -- to the highest -- vertex A -- so that the last write wins)"""
    ab, ac = (B - A)[None], (C - A)[None]
    ap = p[:, None, :] - A[None]
    bp, cp = ap - ab, ap - ac
    dot = lambda x, y: np.einsum("ntk,ntk->nt", np.broadcast_to(x, ap.shape), np.broadcast_to(y, ap.shape))
    d1, d2, d3, d4, d5, d6 = dot(ab, ap), dot(ac, ap), dot(ab, bp), dot(ac, bp), dot(ab, cp), dot(ac, cp)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        den = 1.0 / (va + vb + vc)
        v, w = vb * den, vc * den
        e43, e56 = d4 - d3, d5 - d6
        m = (va <= 0) & (e43 >= 0) & (e56 >= 0); wbc = e43 / (e43 + e56); v = np.where(m, 1 - wbc, v); w = np.where(m, wbc, w)
        m = (vb <= 0) & (d2 >= 0) & (d6 <= 0); v = np.where(m, 0.0, v); w = np.where(m, d2 / (d2 - d6), w)
        m = (d6 >= 0) & (d5 <= d6); v = np.where(m, 0.0, v); w = np.where(m, 1.0, w)
        m = (vc <= 0) & (d1 >= 0) & (d3 <= 0); v = np.where(m, d1 / (d1 - d3), v); w = np.where(m, 0.0, w)
        m = (d3 >= 0) & (d4 <= d3); v = np.where(m, 1.0, v); w = np.where(m, 0.0, w)
        m = (d1 <= 0) & (d2 <= 0); v = np.where(m, 0.0, v); w = np.where(m, 0.0, w)
    e = ap - v[..., None] * ab - w[..., None] * ac
    d2_ = np.einsum("ntk,ntk->nt", e, e)
    return np.where(np.isfinite(d2_), d2_, np.inf)


def _sphere_hull_dist(sc, sr, bc, Rb, hull):
    """signed distance of spheres to the convex hull of `hull` [nv, 3] (object frame) -- csrc/pbre_core.hpp: sphere_hull.  Faces from
    scipy's Qhull (any triangulation of the surface gives the same distances)."""
    from scipy.spatial import ConvexHull
    H = ConvexHull(hull)
    dl = np.einsum("nji,nj->ni", Rb, sc - bc)
    sd = (dl @ H.equations[:, :3].T + H.equations[:, 3][None]).max(axis=1)       # largest signed plane distance: <= 0 inside
    tri = hull[H.simplices]
    dist = np.sqrt(_closest_on_triangles(dl, tri[:, 0], tri[:, 1], tri[:, 2]).min(axis=1))
    return np.where(sd <= 0, sd - sr, dist - sr)


def _shape_candidates(shape, h, Ro):
    """[N, 8, 3] candidate contact points of the object against its support (offsets, world axes) and [N, 8] validity"""
    n = Ro.shape[0]
    r = np.zeros((n, 8, 3)); ok = np.zeros((n, 8), bool)
    for v in range(8):
        if shape == 0:
            l = np.array([h[0] if v & 1 else -h[0], h[1] if v & 2 else -h[1], h[2] if v & 4 else -h[2]])
            r[:, v] = np.einsum("nij,j->ni", Ro, l); ok[:, v] = True
        elif shape == 1:
            if v == 0:
                r[:, v] = [0.0, 0.0, -h[0]]; ok[:, v] = True
        else:
            s, k = (-1.0 if v < 4 else 1.0), v & 3
            if k < 3:
                l = np.array([h[0] * np.cos(2 * np.pi * k / 3), h[0] * np.sin(2 * np.pi * k / 3), s * h[2]])
                r[:, v] = np.einsum("nij,j->ni", Ro, l); ok[:, v] = True
            else:
                dx, dy = -Ro[:, 2, 0], -Ro[:, 2, 1]
                ln = np.hypot(dx, dy)
                good = ln >= 1e-6
                l = np.stack([h[0] * dx / np.maximum(ln, 1e-30), h[0] * dy / np.maximum(ln, 1e-30), np.full(n, s * h[2])], 1)
                r[:, v] = np.einsum("nij,nj->ni", Ro, l); ok[:, v] = good
    return r, ok


def contact_flags(table, state, ndof, phys, no_object=False, hull=None):
    """[N] uint8 of OBJECT_TABLE | ROBOT_OBJECT | ROBOT_TABLE for the batch state records state[N, F] (Q | V | X layout of
    include/pbre.h: joints at [0, ndof), object position / quaternion behind them); phys = pbre_physics (Engine.get_physics()).
    hull: [nv, 3] vertices of a convex-hull object (phys.obj_shape 3: the vertex set handed to Engine.set_object_hull)."""
    table = np.asarray(table, float)
    st = np.asarray(state, float)
    n = st.shape[0]
    nl, ns = int(table[2]), int(table[5])
    R, p = link_frames(table, st[:, :int(table[3])])      # (ndof: where the object sits in the record = Engine.obj_off; the joints are the table's)
    margin = float(phys.contact_margin)
    tc, th = np.array(list(phys.table_c), float), np.array(list(phys.table_h), float)
    oh = np.array(list(phys.obj_h), float)
    shape = int(getattr(phys, "obj_shape", 0))
    op, Ro = st[:, ndof:ndof + 3], _quat_R(st[:, ndof + 3:ndof + 7])
    eye = np.broadcast_to(np.eye(3), (n, 3, 3))
    flags = np.zeros(n, np.uint8)
    base = HEADER + nl * LINK_STRIDE
    for k in range(ns):
        s = table[base + k * SPHERE_STRIDE: base + (k + 1) * SPHERE_STRIDE]
        li, c, rad = int(s[0]), s[1:4], float(s[4])
        sc = p[:, li] + np.einsum("nij,j->ni", R[:, li], c)
        if not no_object:
            if shape == 3:
                if hull is None:
                    raise ValueError("contact_flags: a convex-hull object needs its vertices (hull=)")
                d_ro = _sphere_hull_dist(sc, rad, op, Ro, np.asarray(hull, float))
            else:
                d_ro = _sphere_box_dist(sc, rad, op, Ro, oh) if shape == 0 else _sphere_round_dist(shape, sc, rad, op, Ro, oh)
            flags |= np.where(d_ro < margin, ROBOT_OBJECT, 0).astype(np.uint8)
        flags |= np.where(_sphere_box_dist(sc, rad, np.broadcast_to(tc, (n, 3)), eye, th) < margin, ROBOT_TABLE, 0).astype(np.uint8)
    if not no_object:                     # the object's candidate points (box vertices / round primitives) against the table top
        top, bot = tc[2] + th[2], tc[2] - th[2]
        if shape == 3:
            cand = np.einsum("nij,vj->nvi", Ro, np.asarray(hull, float)); ok = np.ones(cand.shape[:2], bool)      # every vertex is a candidate
        else:
            cand, ok = _shape_candidates(shape, oh, Ro)
        for v in range(cand.shape[1]):
            x = op + cand[:, v]
            on = (np.abs(x[:, 0] - tc[0]) <= th[0]) & (np.abs(x[:, 1] - tc[1]) <= th[1]) & (x[:, 2] > bot) & ok[:, v]
            flags |= np.where(on & (x[:, 2] - top < margin), OBJECT_TABLE, 0).astype(np.uint8)
    return flags
