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
Lateral friction 1.0 (cube_small.urdf) for the pybullet_data objects, PyBullet's default 0.5 for the YCB URDFs [EXT-UNVERIFIED]."""

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
SHAPE_BOX, SHAPE_SPHERE, SHAPE_CYLINDER = 0, 1, 2


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
