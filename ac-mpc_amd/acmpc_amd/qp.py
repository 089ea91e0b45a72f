"""Small dense operator-splitting QP solver for the host-side speed profile.

    minimise 1/2 x'Px + q'x   subject to   l <= A x <= u          (P diagonal, positive)

The reference hands this problem to the third-party `osqp` package (speed_profile.py:61-86), which is not a
dependency of this build.  The iteration below is the ADMM splitting published by Stellato, Banjac, Goulart,
Bemporad and Boyd ("OSQP: an operator splitting solver for quadratic programs", 2020) in its plainest form - one
Cholesky factorisation per solve, over-relaxation, per-row step sizes with a heavier weight on equality rows, a
residual-balancing restart of the step size - written for NumPy and for the tiny dense systems of this controller
(n = horizon - 1 ~ 50..100).  Termination uses the same absolute/relative residual test and the same default
tolerances (1e-3) as the package the reference calls, so "solved" means the same thing to the caller.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
from scipy.linalg import cho_factor, cho_solve

SOLVED = "solved"
MAX_ITER = "maximum iterations reached"


def solve_qp(P_diag, q, A, l, u, *, max_iter=4000, eps_abs=1e-3, eps_rel=1e-3, rho=0.1, sigma=1e-6, alpha=1.6,
             x0=None, y0=None, check_every=10, adapt_every=50) -> SimpleNamespace:
    """Returns a namespace shaped like osqp's result: `.x`, `.y`, `.info.status`, `.info.iter`."""
    P_diag = np.asarray(P_diag, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    A = np.asarray(A, dtype=np.float64)
    l = np.asarray(l, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    m, n = A.shape
    eq = l == u
    At = A.T

    def factor(rho_now):
        rho_rows = np.where(eq, 1e3 * rho_now, rho_now)
        K = At @ (rho_rows[:, None] * A)
        K[np.diag_indices(n)] += P_diag + sigma
        return rho_rows, cho_factor(K, lower=True, check_finite=False)

    rho_rows, chol = factor(rho)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    y = np.zeros(m) if y0 is None else np.array(y0, dtype=np.float64)
    z = np.clip(A @ x, l, u)
    status, it = MAX_ITER, 0
    for it in range(1, max_iter + 1):
        x_tilde = cho_solve(chol, sigma * x - q + At @ (rho_rows * z - y), check_finite=False)
        z_tilde = A @ x_tilde
        x = alpha * x_tilde + (1.0 - alpha) * x
        z_mix = alpha * z_tilde + (1.0 - alpha) * z
        z_new = np.clip(z_mix + y / rho_rows, l, u)
        y = y + rho_rows * (z_mix - z_new)
        z = z_new
        if it % check_every:
            continue
        Ax = A @ x
        Aty = At @ y
        Px = P_diag * x
        r_prim = np.abs(Ax - z).max()
        r_dual = np.abs(Px + q + Aty).max()
        s_prim = max(np.abs(Ax).max(), np.abs(z).max())
        s_dual = max(np.abs(Px).max(), np.abs(Aty).max(), np.abs(q).max())
        if r_prim <= eps_abs + eps_rel * s_prim and r_dual <= eps_abs + eps_rel * s_dual:
            status = SOLVED
            break
        if it % adapt_every == 0:
            ratio = np.sqrt((r_prim / max(s_prim, 1e-12)) / max(r_dual / max(s_dual, 1e-12), 1e-12))
            if ratio > 5.0 or ratio < 0.2:
                rho = float(np.clip(rho * ratio, 1e-6, 1e6))
                rho_rows, chol = factor(rho)
    return SimpleNamespace(x=x, y=y, info=SimpleNamespace(status=status, iter=it))
