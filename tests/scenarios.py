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
        q = np.clip(q, lo + 1e-3, hi - 1e-3)     # stay off the exact limit: fp32 rounding of the limit would flip the limit row
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
        q = np.clip(q, lo + 1e-3, hi - 1e-3)
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


def ik_position(oracle, target, q0=None, iters=200):
    """Damped least-squares IK on the arm joints for the EE position (test helper only)."""
    q = (HOME if q0 is None else q0).copy()
    ee = oracle.model.ee_link
    lo, hi = joint_limits(oracle)
    for _ in range(iters):
        R, p = oracle.fk(q)
        e = np.asarray(target) - p[ee]
        if np.linalg.norm(e) < 1e-5:
            break
        J = np.zeros((3, 7))
        for j in range(7):
            dq = q.copy(); dq[j] += 1e-6
            J[:, j] = (oracle.fk(dq)[1][ee] - p[ee]) / 1e-6
        q[:7] += J.T @ np.linalg.solve(J @ J.T + 1e-4 * np.eye(3), e) * 0.5
        q = np.clip(q, lo, hi)
    return q


def push_actions(oracle, state, steps_approach=120, steps_push=160):
    """Open-loop action sequence: move the hand behind the cube (-x side), then sweep +x through it."""
    obj = np.asarray(state[9:12], dtype=float)
    q_pre = ik_position(oracle, obj + [-0.10, 0.0, 0.035])
    q_end = ik_position(oracle, obj + [0.08, 0.0, 0.035], q0=q_pre)
    return q_pre, q_end, steps_approach, steps_push


def track(q, q_goal, amax=1.0):
    """Action that moves the arm joints toward q_goal (0.05 rad per unit action, position gain 0.5 ->
    0.025 rad/step at |a| = 1); amax limits the speed."""
    return np.clip((q_goal[:7] - q[:7]) / 0.05 * 2.0, -amax, amax)
