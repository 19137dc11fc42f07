"""Stand-in geometry of the objects the reference loads by name (reference world_env.py:18-25, 61-84, 179-216).

The reference's objects are meshes of packages that are not part of its checkout (`pybullet_data`: duck_vhacd, teddy_vhacd,
domino, cube_small; `pybullet_object_models`: the YCB set and its superquadric approximations) and not available to this build.
Every `obj_name` is simulated as a collision PRIMITIVE with the object's approximate dimensions and mass -- a box, or for the round
members (ROUND_OBJECTS: tennis ball, pear, strawberry, the cans, the duck) a sphere / an upright cylinder -- so that `obj_name` changes
the dynamics (shape, size, mass, principal inertias, lateral friction) instead of being ignored:
  * YCB objects: dimensions (m) and masses (kg) as published with the YCB object and model set (Calli et al., "Benchmarking in
    Manipulation Research", 2015, object table); box axes = the published x, y, z extents;
  * pybullet_data objects: approximate extents of the meshes at the scale their URDFs load them with [EXT-UNVERIFIED: the package is
    absent; cube_small is the 5 cm / 0.1 kg cube of SURVEY Appendix B].
Lateral friction 1.0 (cube_small.urdf) for the pybullet_data objects, PyBullet's default 0.5 for the YCB URDFs [EXT-UNVERIFIED].

Round 6: where the object's MESH can be read -- `pybullet_data` / `pybullet_object_models` importable, or a directory of
`<obj_name>.obj` files named by PBRE_OBJECT_MESH_DIR -- `object_physics` returns the object as a CONVEX HULL instead (obj_shape 3, at most 32
vertices, mass properties of the uniform solid scaled to the table's mass): `hull_physics` / `find_mesh`.  The engine's narrow phase for it
is include/pbre.h: pbre_set_object_hull.  Bullet collides each piece of a *_vhacd decomposition as a convex hull of its own; one hull of the
whole mesh is its convex envelope [a documented deviation; the dominant contact geometry of a duck or a can on a table is its envelope]."""

# name -> (full extents x, y, z [m], mass [kg], lateral friction)
PYBULLET_DATA_OBJECTS = {
    "cube_small": ((0.05, 0.05, 0.05), 0.1, 1.0),
    "duck_vhacd": ((0.09, 0.07, 0.08), 0.1, 1.0),
    "teddy_vhacd": ((0.10, 0.08, 0.12), 0.1, 1.0),
    "domino/domino": ((0.048, 0.024, 0.008), 0.02, 1.0),
    "lego/lego": ((0.032, 0.024, 0.05), 0.1, 1.0),
}
YCB_OBJECTS = {
    "YcbBanana": ((0.19, 0.036, 0.036), 0.066, 0.5),
    "YcbChipsCan": ((0.075, 0.075, 0.25), 0.205, 0.5),
    "YcbCrackerBox": ((0.06, 0.158, 0.21), 0.411, 0.5),
    "YcbFoamBrick": ((0.05, 0.075, 0.05), 0.028, 0.5),
    "YcbGelatinBox": ((0.028, 0.085, 0.073), 0.097, 0.5),
    "YcbHammer": ((0.024, 0.032, 0.135), 0.665, 0.5),
    "YcbMasterChefCan": ((0.102, 0.102, 0.139), 0.414, 0.5),
    "YcbMediumClamp": ((0.09, 0.115, 0.027), 0.059, 0.5),
    "YcbMustardBottle": ((0.058, 0.095, 0.19), 0.603, 0.5),
    "YcbPear": ((0.066, 0.066, 0.10), 0.049, 0.5),
    "YcbPottedMeatCan": ((0.05, 0.097, 0.082), 0.37, 0.5),
    "YcbPowerDrill": ((0.035, 0.046, 0.184), 0.895, 0.5),
    "YcbScissors": ((0.087, 0.20, 0.014), 0.082, 0.5),
    "YcbStrawberry": ((0.044, 0.044, 0.055), 0.018, 0.5),
    "YcbSugarBox": ((0.038, 0.089, 0.175), 0.514, 0.5),
    "YcbTennisBall": ((0.065, 0.065, 0.065), 0.058, 0.5),
    "YcbTomatoSoupCan": ((0.066, 0.066, 0.101), 0.349, 0.5),
}


# Round members of the object list (reference world_env.py:18-25, 179-216): collision primitive instead of the bounding box, so that they
# roll where the reference's meshes roll.  name -> ("sphere", radius) | ("cylinder", radius, half height); the cylinder's axis is the
# object's z (the cans stand upright after reset, as their URDFs load them).  Radii / heights from the same published extents;
# a sphere that stands in for an ellipsoid (pear, strawberry) has the volume-equivalent radius.  [EXT-UNVERIFIED like the boxes.]
ROUND_OBJECTS = {
    "YcbTennisBall": ("sphere", 0.0325),
    "YcbPear": ("sphere", 0.038),
    "YcbStrawberry": ("sphere", 0.024),
    "YcbChipsCan": ("cylinder", 0.0375, 0.125),
    "YcbMasterChefCan": ("cylinder", 0.051, 0.0695),
    "YcbTomatoSoupCan": ("cylinder", 0.033, 0.0505),
    "duck_vhacd": ("cylinder", 0.04, 0.04),        # a rounded body on a flat base: slides on its base, rolls on its side
}
SHAPE_BOX, SHAPE_SPHERE, SHAPE_CYLINDER, SHAPE_HULL = 0, 1, 2, 3
HULL_MAXV = 32      # include/pbre.h PBRE_HULL_MAXV


def object_physics(obj_name):
    """pbre_physics fields of the stand-in: collision primitive (obj_shape), its dimensions (obj_h: half extents | radius x 3 | radius,
    radius, half height), mass, principal inertias of the uniform solid, lateral friction."""
    key = obj_name[:-5] if obj_name.endswith(".urdf") else obj_name
    ent = PYBULLET_DATA_OBJECTS.get(key) or YCB_OBJECTS.get(key)
    if ent is None:
        raise ValueError("unknown obj_name %r; known: %s" % (obj_name, sorted(list(PYBULLET_DATA_OBJECTS) + list(YCB_OBJECTS))))
    (x, y, z), mass, mu = ent
    rnd = ROUND_OBJECTS.get(key)
    if rnd is None:
        return {"obj_shape": SHAPE_BOX, "obj_h": [x / 2, y / 2, z / 2], "obj_mass": mass, "obj_mu": mu,
                "obj_inertia": [mass * (y * y + z * z) / 12.0, mass * (x * x + z * z) / 12.0, mass * (x * x + y * y) / 12.0]}
    if rnd[0] == "sphere":
        r = rnd[1]
        return {"obj_shape": SHAPE_SPHERE, "obj_h": [r, r, r], "obj_mass": mass, "obj_mu": mu, "obj_inertia": [0.4 * mass * r * r] * 3}
    r, hh = rnd[1], rnd[2]
    it = mass * (3.0 * r * r + 4.0 * hh * hh) / 12.0
    return {"obj_shape": SHAPE_CYLINDER, "obj_h": [r, r, hh], "obj_mass": mass, "obj_mu": mu, "obj_inertia": [it, it, 0.5 * mass * r * r]}


# ---------------------------------------------------------------------------------------------------------------- convex-hull objects
def read_obj_vertices(path):
    """the `v x y z` records of a Wavefront .obj file -> [n, 3] float64"""
    import numpy as np
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                t = line.split()
                out.append([float(t[1]), float(t[2]), float(t[3])])
    if len(out) < 4:
        raise ValueError("%s: fewer than 4 vertices" % path)
    return np.asarray(out, np.float64)


def _hull_mass_properties(pts, simplices):
    """volume, centroid and inertia tensor about the centroid (unit density) of the closed triangle surface `simplices` over `pts`,
    by signed tetrahedra against the origin (orientation fixed per triangle so that every tetrahedron counts positive for a convex body
    around an interior point)."""
    import numpy as np
    c0 = pts.mean(0)
    P = pts - c0
    vol, cen = 0.0, np.zeros(3)
    C = np.zeros((3, 3))                     # covariance integral  int x x^T dV
    canon = np.array([[2, 1, 1], [1, 2, 1], [1, 1, 2]], float) / 120.0
    for tri in simplices:
        a, b, c = P[tri[0]], P[tri[1]], P[tri[2]]
        det = float(np.dot(a, np.cross(b, c)))
        if det < 0:
            b, c = c, b
            det = -det
        A = np.stack([a, b, c], 1)           # columns
        vol += det / 6.0
        cen += det / 24.0 * (a + b + c)
        C += det * A @ canon @ A.T
    cen /= vol
    C -= vol * np.outer(cen, cen)
    I = np.trace(C) * np.eye(3) - C
    return vol, cen + c0, I


def reduce_vertices(pts, max_v=HULL_MAXV):
    """at most max_v of the hull vertices of `pts`: all of them if they are few enough, else the extreme points along the coordinate axes
    followed by farthest-point sampling (every kept point is a vertex of the original hull, so the reduced hull lies inside it)."""
    import numpy as np
    from scipy.spatial import ConvexHull
    hv = pts[ConvexHull(pts).vertices]
    if len(hv) <= max_v:
        return hv
    keep = []
    for k in range(3):
        for i in (int(np.argmin(hv[:, k])), int(np.argmax(hv[:, k]))):
            if i not in keep:
                keep.append(i)
    d = np.min(np.linalg.norm(hv[:, None, :] - hv[None, keep, :], axis=2), axis=1)
    while len(keep) < max_v:
        i = int(np.argmax(d))
        keep.append(i)
        d = np.minimum(d, np.linalg.norm(hv - hv[i], axis=1))
    return hv[sorted(keep)]


def hull_physics(vertices, mass, mu, scale=1.0):
    """pbre_physics fields + `obj_hull` of the convex hull of `vertices` ([n, 3]; a mesh's vertex list): vertices relative to the centre
    of mass of the uniform solid, obj_h = half extents of their bounding box about it, obj_inertia = the diagonal of the solid's inertia
    tensor in the mesh's own axes (the engine's object frame keeps the mesh's axes -- what the object's observed Euler angles refer to --,
    so products of inertia are dropped: exact for the symmetric objects, an approximation otherwise)."""
    import numpy as np
    from scipy.spatial import ConvexHull
    v = reduce_vertices(np.asarray(vertices, np.float64) * float(scale))
    h = ConvexHull(v)
    vol, cen, I = _hull_mass_properties(v, h.simplices)
    v = v[np.sort(h.vertices)] - cen
    dens = mass / vol
    return {"obj_shape": SHAPE_HULL, "obj_hull": v, "obj_h": [float(np.abs(v[:, k]).max()) for k in range(3)], "obj_mass": mass, "obj_mu": mu,
            "obj_inertia": [float(dens * I[k, k]) for k in range(3)]}


def find_mesh(obj_name):
    """path of the collision mesh of `obj_name`, or None: PBRE_OBJECT_MESH_DIR/<name>.obj, then the packages the reference loads its objects
    from (world_env.py:14-15, 61-84, 179-216) when they are importable"""
    import os
    key = obj_name[:-5] if obj_name.endswith(".urdf") else obj_name
    base = os.path.basename(key)
    cands = []
    d = os.environ.get("PBRE_OBJECT_MESH_DIR")
    if d:
        cands += [os.path.join(d, base + ".obj"), os.path.join(d, key + ".obj")]
    try:
        import pybullet_data
        root = pybullet_data.getDataPath()
        cands += [os.path.join(root, key + ".obj"), os.path.join(root, base + ".obj")]
    except Exception:
        pass
    try:
        from pybullet_object_models import ycb_objects
        root = ycb_objects.getDataPath()
        cands += [os.path.join(root, key, "collision_vhacd.obj"), os.path.join(root, key, "textured_simple_reoriented.obj")]
    except Exception:
        pass
    for c in cands:
        if os.path.isfile(c):
            return c
    return None


_primitive_physics = object_physics


def object_physics(obj_name, use_mesh=True):      # noqa: F811  (wraps the table lookup above)
    """the object's pbre_physics fields: its mesh's convex hull where the mesh can be read (find_mesh), the primitive stand-in otherwise"""
    prim = _primitive_physics(obj_name)
    path = find_mesh(obj_name) if use_mesh else None
    if path is None or prim["obj_shape"] == SHAPE_BOX and (obj_name[:-5] if obj_name.endswith(".urdf") else obj_name) == "cube_small":
        return prim
    try:
        return hull_physics(read_obj_vertices(path), prim["obj_mass"], prim["obj_mu"])
    except Exception as e:      # an unreadable mesh must not take the env down: the stand-in is the documented fallback
        import warnings
        warnings.warn("object %r: mesh %s not usable (%s); using the primitive stand-in" % (obj_name, path, e))
        return prim
