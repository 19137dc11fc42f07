"""Crafted simulator states that exercise every contact category (tests only)."""
import numpy as np

HOME = np.array([0, -0.54, 0, -2.6, -0.30, 2.0, 1.0, 0.02, 0.02])


def sphere_centres(oracle, model, spheres, q):
    names = [l["name"] for l in model["links"]]
    R, p = oracle.fk(q)
    return [(p[names.index(ln)] + R[names.index(ln)] @ np.array(c), r) for (ln, c, r) in spheres]


def joint_limits(oracle):
    m = oracle.model
    lo = np.array([m.lower[m.link_of_dof[k]] for k in range(oracle.ndof)])
    hi = np.array([m.upper[m.link_of_dof[k]] for k in range(oracle.ndof)])
    return lo, hi


def table_contact_states(oracle, model, spheres, base, n, rng, vel=0.5):
    """Arm configurations with at least one collision sphere within [-4 mm, +0.8 mm] of the table top."""
    lo, hi = joint_limits(oracle)
    out = []
    while len(out) < n:
        q = HOME + rng.normal(0, 0.5, 9)
        q[7:] = rng.uniform(0.0, 0.04, 2)
        q = np.clip(q, lo, hi)
        ds = [c[2] - r - 0.625 for c, r in sphere_centres(oracle, model, spheres, q)
              if 0.1 < c[0] < 1.6 and abs(c[1]) < 0.5]
        if ds and -0.004 < min(ds) < 0.0008:
            s = base.copy()
            s[:9] = q
            s[16:25] = rng.normal(0, vel, 9)
            out.append(s)
    return np.array(out)


def object_contact_states(oracle, model, spheres, base, n, rng, pen=0.002, vel=0.3, rotate=True):
    """Object placed so that one of the hand/finger spheres penetrates it by about `pen`."""
    lo, hi = joint_limits(oracle)
    out = []
    for _ in range(n):
        q = HOME + rng.normal(0, 0.3, 9)
        q[7:] = 0.02
        q = np.clip(q, lo, hi)
        c, r = sphere_centres(oracle, model, spheres, q)[rng.integers(0, 7)]
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        s = base.copy()
        s[:9] = q
        s[16:25] = rng.normal(0, vel, 9)
        s[9:12] = c - d * (r + 0.025 - pen)
        if rotate:
            ang = rng.uniform(-1, 1)
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            s[12:15] = ax * np.sin(ang / 2)
            s[15] = np.cos(ang / 2)
        s[25:31] = rng.normal(0, vel * 0.5, 6)
        out.append(s)
    return np.array(out)
