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
    while len(out) < n:
        q = HOME + rng.normal(0, 0.3, 9)
        q[7:] = 0.02
        q = np.clip(q, lo + 1e-3, hi - 1e-3)
        cs_all = sphere_centres(oracle, model, spheres, q)
        c, r = cs_all[rng.integers(0, 7)]
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
        # (round 4, four robot-object slots: a placement that buries some OTHER sphere centimetres deep in the cube is a garbage state -- with
        # two slots the third such contact was simply dropped, with four it is solved, stiffly.  Like the iCub generator: keep a state
        # only if the oracle lists a robot-object contact and none of them is deeper than 4 mm.)
        _, info = oracle.sim_step(s, q.copy(), np.full(9, 0.1), np.full(9, 1.0))
        ro = [info.dist[k] for k in range(info.ncontacts) if info.type[k] == 1]
        if not ro or min(ro) < -0.004:
            continue
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


def multi_sphere_object_states(oracle, model, spheres, base, n, rng, want=3, vel=0.2, hand_too=0.5):
    """The cube PINCHED between the Panda's fingers: states in which at least `want` (3 or 4) collision spheres lie within the contact
    margin of the object at once ("<= 4 cube-robot points"; the round-3 engine kept the two deepest).  Laterally the fat hand
    spheres shadow the finger spheres, so the only way to more than two contacts on a 5 cm cube is the grasp geometry: both spheres of
    the left finger on one face, both of the right finger on the opposite face (finger opening = cube width + the spheres' offset), and
    -- with probability `hand_too` -- the cube pushed back until the palm sphere touches a third face: five candidates for four
    slots.  Depths vary by a few tenths of a millimetre around the margin.  Returns the states that the oracle confirms to have
    >= `want` robot-object contacts, none deeper than 2 mm."""
    lo, hi = joint_limits(oracle)
    names = [l["name"] for l in model["links"]]
    hand = names.index("panda_hand")
    out = []
    tries = 0
    while len(out) < n and tries < 4000:
        tries += 1
        q = HOME + rng.normal(0, 0.3, 9)
        pen = rng.uniform(-0.0003, 0.0012, 2)                           # per finger: > 0 penetrating, < 0 inside the margin only
        q[7:] = 0.027 - pen
        q = np.clip(q, lo + 1e-3, hi - 1e-3)
        R, p = oracle.fk(q)
        Rh, ph = R[hand], p[hand]
        # (palm sphere: centre z = 0.025, radius 0.04 in the hand frame -> it reaches z = 0.065; the cube's back face is at zc - 0.025)
        zc = 0.090 - rng.uniform(-0.0006, 0.0012) if rng.random() < hand_too else 0.096 + rng.uniform(-0.003, 0.003)
        yc = 0.5 * (pen[1] - pen[0]) * 0.0                              # (finger spheres are symmetric: the cube stays centred)
        centre = ph + Rh @ np.array([rng.uniform(-0.004, 0.004), yc, zc])
        if centre[2] - 0.0433 < 0.64:                                   # keep the cube clear of the table: no object-table rows in this family
            continue
        # cube axes = hand axes, turned by a few milliradians (so that no two spheres are EXACTLY equally deep)
        ax_, ay_, az_ = rng.uniform(-0.012, 0.012, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(ax_), -np.sin(ax_)], [0, np.sin(ax_), np.cos(ax_)]])
        Ry = np.array([[np.cos(ay_), 0, np.sin(ay_)], [0, 1, 0], [-np.sin(ay_), 0, np.cos(ay_)]])
        Rz = np.array([[np.cos(az_), -np.sin(az_), 0], [np.sin(az_), np.cos(az_), 0], [0, 0, 1]])
        Rc = Rh @ Rx @ Ry @ Rz
        tr = np.trace(Rc)
        if tr > 0:
            s4 = np.sqrt(tr + 1.0) * 2; qw = 0.25 * s4; qx = (Rc[2, 1] - Rc[1, 2]) / s4; qy = (Rc[0, 2] - Rc[2, 0]) / s4; qz = (Rc[1, 0] - Rc[0, 1]) / s4
        else:
            ii = int(np.argmax(np.diag(Rc))); jj, kk = (ii + 1) % 3, (ii + 2) % 3
            s4 = np.sqrt(1.0 + Rc[ii, ii] - Rc[jj, jj] - Rc[kk, kk]) * 2
            v = [0.0, 0.0, 0.0]; v[ii] = 0.25 * s4; v[jj] = (Rc[jj, ii] + Rc[ii, jj]) / s4; v[kk] = (Rc[kk, ii] + Rc[ii, kk]) / s4
            qx, qy, qz = v; qw = (Rc[kk, jj] - Rc[jj, kk]) / s4
        s = base.copy()
        s[:9] = q
        s[16:25] = rng.normal(0, vel, 9)
        s[9:12] = centre
        s[12:16] = [qx, qy, qz, qw]
        s[25:31] = rng.normal(0, vel * 0.3, 6)
        _, info = oracle.sim_step(s, q.copy(), np.full(9, 0.1), np.full(9, 1.0))
        ro = [info.dist[c] for c in range(info.ncontacts) if info.type[c] == 1]
        if len(ro) >= want and min(ro) > -0.002:
            out.append(s)
    return np.array(out)
