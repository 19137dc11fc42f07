"""The validity bound of the object block's closed form (csrc/pbre_fast.hpp: Fast::obj_closed), checked independently of
the engine in fp64 numpy (tests/objblock_ref.py): over thousands of random contact configurations -- resting, sliding near the edge of
the friction cone, rocking, lifting off, 3 and 2 contacts -- whenever the bound holds after the explicit sweeps, NO clamp binds in any
of the remaining sweeps of Bullet's sequential rows and the matrix power equals their result; and the bound is not vacuous (it holds
for the resting cubes, it fails for the sliding / rocking ones)."""
import numpy as np

import objblock_ref as ob

K, N = 22, 128


def random_case(rng, kind):
    sk, h = 49.0, 0.025                      # sqrt(m / I) of the 5 cm cube, half extent
    yaw = rng.uniform(0, 2 * np.pi)
    cy, sy = np.cos(yaw), np.sin(yaw)
    tilt = 0.0 if kind in ("rest", "slide", "press") else 10 ** rng.uniform(-4, -1.2)
    r = []
    for sx_, sy_ in ((-1, -1), (1, -1), (-1, 1), (1, 1)):
        lx, ly = sx_ * h, sy_ * h
        r.append([sk * (cy * lx - sy * ly), sk * (sy * lx + cy * ly), sk * (-h + tilt * lx)])
    r = np.array(r)
    J = ob.rows_of(r)
    active = np.ones(4, bool)
    if kind == "three":
        active[rng.integers(4)] = False
    if kind == "two":
        active[[0, 1]] = False
    dt, g = 1 / 240.0, 9.8
    x = np.zeros(6); x[2] = -g * dt
    beta = np.zeros(12)
    # penetration (+ slop) of the four vertices: a rigid flat face gives values affine in the vertex position (a consistent system);
    # the rocking / lifting cases get independent ones (some separating: pen > 0 switches the row's right-hand side)
    if kind in ("rest", "slide", "press", "three", "two"):
        a0, a1, a2 = -10 ** rng.uniform(-5.5, -4), rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-4, 2e-4)
        pen = np.array([a0 + a1 * sx_ * h + a2 * sy_ * h for sx_, sy_ in ((-1, -1), (1, -1), (-1, 1), (1, 1))])
    else:
        pen = rng.uniform(-2e-4, 1e-5, 4)
    for c in range(4):
        dinv = 1.0 / (J[c] @ J[c])
        beta[c] = (-pen[c] / dt if pen[c] > 0 else -pen[c] * 0.2 / dt) * dinv
    if kind == "slide":
        v = 10 ** rng.uniform(-5, -0.5); a = rng.uniform(0, 2 * np.pi)
        x[0], x[1] = v * np.cos(a), v * np.sin(a)
    if kind == "rock":
        x[3:5] = rng.normal(0, 1, 2) * 10 ** rng.uniform(-3, 0) / sk
    if kind == "lift":
        x[2] = 10 ** rng.uniform(-4, -1)
    if kind == "press":
        x[2] = -10 ** rng.uniform(-3, 0)
    mu = rng.choice([0.5, 0.3, 1.0])
    return J, beta, mu, x, active


def test_bound_implies_no_clamp_in_the_remaining_sweeps():
    rng = np.random.default_rng(2024)
    held = {}
    for kind in ("rest", "slide", "rock", "lift", "press", "three", "two"):
        held[kind] = [0, 0]
        for _ in range(250):
            J, beta, mu, x, active = random_case(rng, kind)
            app = np.zeros(12)
            for _k in range(K):
                ob.sweep(J, beta, mu, x, app, active)
            xt, ok = ob.bound_holds(J, beta, mu, x.copy(), app.copy(), active, N)
            held[kind][1 if ok else 0] += 1
            if not ok:
                continue
            rec = []
            for _k in range(N):
                ob.sweep(J, beta, mu, x, app, active, rec)
            assert not rec, (kind, rec[:4])
            assert np.abs(x - xt).max() <= 1e-9 * max(1.0, np.abs(x).max()), (kind, np.abs(x - xt).max())
    print(held)
    assert held["rest"][1] > 200, held                       # the resting cubes take the closed form (the rest: a vertex just outside the slop)
    assert held["press"][1] > 200, held
    assert held["slide"][0] > 50 and held["slide"][1] > 20 and held["rock"][0] > 20 and held["lift"][0] > 20, held      # ... and the bound does reject


def test_impulse_movement_bound_with_small_sigma_and_large_residuals():
    """the inequality the test above rests on -- over n UNCLAMPED sweeps a row's applied impulse moves by at most
    n |eta_r| + T from its value after the explicit sweeps -- checked directly where the old T = (16 (rho + 17 E) + 17 n E) / (1 - sigma)
    was too small: well-conditioned blocks (sigma near 0) with INCONSISTENT right-hand sides (large E: the rows' deltas at the closed form's
    result do not vanish, each sweep moves every impulse by about eta_r for ever)."""
    rng = np.random.default_rng(77)
    worst = 0.0
    for case in range(300):
        # short lever arms make the 12 rows nearly orthogonal triples: S^16 is tiny
        r = rng.normal(0, 1, (4, 3)) * 10 ** rng.uniform(-1.5, 0.5)
        J = ob.rows_of(r)
        active = np.ones(4, bool)
        beta = rng.normal(0, 1, 12) * 10 ** rng.uniform(-3, 0)          # friction rows with a right-hand side too: as inconsistent as it gets
        beta[4:] *= rng.choice([0.0, 1.0])
        x = rng.normal(0, 1, 6)
        free = lambda xx, aa: _free_sweep(J, beta, xx, aa)
        app = np.zeros(12)
        for _k in range(K):
            free(x, app)
        xt, _ok, det = ob.bound_holds(J, beta, 1.0, x.copy(), app.copy(), active, N, details=True)
        if det is None:
            continue
        eta, T, sig, E = det
        moved = np.zeros(12)
        for _k in range(N):
            before = app.copy()
            free(x, app)
            moved += np.abs(app - before)
        bound = N * eta + T
        assert (moved <= bound * (1 + 1e-9) + 1e-12).all(), (case, sig, E, (moved / np.maximum(bound, 1e-300)).max())
        worst = max(worst, (moved / np.maximum(bound, 1e-300)).max())
    print("largest moved / bound:", worst)
    assert worst > 0.02, "the cases do exercise the bound (measured: 0.046 -- the triangle inequalities over 12 rows x 128 sweeps are not tight)"


def _free_sweep(J, beta, x, app):
    for i in range(12):
        dinv = 1.0 / (J[i] @ J[i])
        d = beta[i] - dinv * (J[i] @ x)
        app[i] += d
        x += d * J[i]
