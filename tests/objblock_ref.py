"""TEST INFRASTRUCTURE: numpy (fp64) restatement of the simple class's object block -- the 12 object-table rows of a resting cube in
the scaled coordinates of csrc/pbre_fast.hpp (step_t: orow; unit mass and unit inertia, impulses in delta-v units) -- as Bullet's
sequential clamped rows, with a record of every clamp that binds, and of the validity bound of Fast::obj_closed.  Used by
tests/test_objblock_bound.py to check the bound's claim independently of the engine: if the bound holds after K sweeps, no clamp
binds in any later sweep."""
import numpy as np


def rows_of(r):
    """r: [4, 3] scaled lever arms -> J [12, 6] in sweep order (4 normals, then the friction pairs), dirs +z, -y, +x"""
    J = np.zeros((12, 6))
    for c in range(4):
        rx, ry, rz = r[c]
        J[c] = [0, 0, 1, ry, -rx, 0]
        J[4 + 2 * c] = [0, -1, 0, rz, 0, -rx]
        J[5 + 2 * c] = [1, 0, 0, 0, rz, -ry]
    return J


def sweep(J, beta, mu, x, app, active, record=None):
    """one sweep of sequential clamped rows (in place).  record: list that receives (row, kind) for every clamp that binds"""
    for i in range(12):
        c = i if i < 4 else (i - 4) // 2
        if not active[c]:
            continue
        dinv = 1.0 / (J[i] @ J[i])
        d = beta[i] - dinv * (J[i] @ x)
        if i < 4:
            s = max(app[i] + d, 0.0)
            if s != app[i] + d and record is not None:
                record.append((i, "normal at 0"))
        else:
            hi = mu * app[c]
            if hi > 0:
                s = min(max(app[i] + d, -hi), hi)
                if s != app[i] + d and record is not None:
                    record.append((i, "friction cone"))
            else:
                s = app[i]
                if record is not None:
                    record.append((i, "no normal impulse"))
        dd = s - app[i]
        app[i] = s
        x += dd * J[i]


def bound_holds(J, beta, mu, x_k, app, active, n, details=False):
    """Fast::obj_closed's test in fp64: (closed-form result, holds?); details: also (eta, T, sigma, E), the pieces of mov_r = n eta_r + T"""
    S = np.eye(6); s = np.zeros(6)
    for i in range(12):
        c = i if i < 4 else (i - 4) // 2
        if not active[c]:
            continue
        dinv = 1.0 / (J[i] @ J[i])
        P = np.eye(6) - dinv * np.outer(J[i], J[i])
        S = P @ S; s = P @ s + beta[i] * J[i]
    A = np.eye(7); A[:6, :6] = S; A[:6, 6] = s
    xt = (np.linalg.matrix_power(A, n) @ np.append(x_k, 1.0))[:6]
    sig = np.linalg.norm(np.linalg.matrix_power(S, 16), "fro")
    rho = np.linalg.norm(x_k - xt)
    eta = np.zeros(12); E = 0.0
    for i in range(12):
        c = i if i < 4 else (i - 4) // 2
        if not active[c]:
            continue
        eta[i] = abs(beta[i] - (J[i] @ xt) / (J[i] @ J[i]))
        E += eta[i] * np.sqrt(1.0 + J[c][3] ** 2 + J[c][4] ** 2 + J[4 + 2 * c][3] ** 2)      # |J_r| <= sqrt(1 + |r'|^2)
    if not sig < 0.9:
        return (xt, False, None) if details else (xt, False)
    T = 16 * (rho + n * E) / (1 - sig) + 8 * n * E
    ok = True
    for c in range(4):
        if not active[c]:
            continue
        nmin = app[c] - 2 * (n * eta[c] + T)
        okc = nmin > 0
        for i in (4 + 2 * c, 5 + 2 * c):
            okc = okc and abs(app[i]) + 2 * (n * eta[i] + T) <= mu * nmin
        ok = ok and okc
    if details:
        return xt, ok, (eta, T, sig, E)
    return xt, ok
