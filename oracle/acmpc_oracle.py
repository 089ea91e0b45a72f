"""CPU oracle for the ac-mpc rollout-and-cost hot path (TEST INFRASTRUCTURE ONLY).

This module restates, in plain NumPy, the arithmetic of the reference controller
(`/root/reference/src/acmpc/...`, cited per function as file:line) plus the
build-defined sampling composition of SURVEY.md section 8 ("mode S" / "mode T").
It is the checker the HIP path is compared against.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product package (`ac-mpc_amd/`) never does.

Parity status
-------------
* Every *ingredient* (waypoints, Frenet transforms, linearisation, QP assembly,
  kinematic step, nearest waypoint, weighted mean, command selection) is pinned
  by golden vectors produced by importing the reference itself
  (`tests/golden/gen_golden.py`).
* The *composition* "sample N control sequences -> roll out -> cost -> argmin"
  does not exist in the reference (its MPC is one OSQP QP), so it is
  BUILD-DEFINED here; it is tied to the reference through the identities
  `A_eq z = l_eq` and `J - const = 1/2 z'Pz + q'z` checked against the
  reference-assembled QP matrices.  OSQP itself is absent from this image
  (un-pinned third-party dependency): QP *solutions* are parity-unpinned.

Numerical specification of the sampling path ("spec order")
-----------------------------------------------------------
The float32 variants below fix one operation order with no library
transcendentals and no implicit fused multiply-add, so that NumPy (here), the C
restatement (`acmpc_oracle.c`) and the HIP kernels (`-ffp-contract=off`) are
bit-identical.  Mode S uses no FMA at all.  Mode T's specification names its
FMAs explicitly (`fma32` below: an exact emulation of IEEE `fmaf` - float64
product, TwoSum, round-to-odd, one rounding to float32 - which C states as
`fmaf` and the kernels as `v_fma_f32` / `v_pk_fma_f32`).
The float64 variants are the same formulas in double precision with libm
trigonometry: they measure the fp32 drift, they are not the parity target.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

EPS = 1e-12  # spatial_mpc.py:34, dynamics.py:21, speed_profile.py:114

# Row order of the 7 x n waypoint table (control/paths.py:4-72)
ROW_X, ROW_Y, ROW_PSI, ROW_KAPPA, ROW_DS, ROW_WIDTH, ROW_V = range(7)

# Width (floats) of one packed coefficient row handed to the kernels
COEF_STRIDE_S = 12
COEF_STRIDE_T = 8

# Mode-S coefficient columns
CS_DS, CS_A21, CS_A31, CS_B31, CS_F3, CS_VREF, CS_KREF, CS_EYLO, CS_EYHI = range(9)
# Mode-T waypoint columns
CT_X, CT_Y, CT_COS, CT_SIN, CT_PSI, CT_KREF, CT_VREF, CT_HALF = range(8)

T_MIN = 0.01  # control.py:134  (x_min = [-inf, -inf, 0.01])
U_SLACK_V = 0.1  # control.py:138-139 (velocity box widened by 0.1)


# ---------------------------------------------------------------------------
# Ingredients pinned by the reference
# ---------------------------------------------------------------------------
def wrap_to_pi(angle):
    """(-pi, pi] wrap used everywhere in the reference (dynamics.py:36, spatial_mpc.py:149-150)."""
    return np.mod(angle + math.pi, 2.0 * math.pi) - math.pi


def construct_waypoints(coords: np.ndarray) -> np.ndarray:
    """H x 3 `[x, y, width]` -> 7 x n table, n = H-1 (spatial_mpc.py:125-154).

    psi/ds look ahead to the next point, width is taken from the *next* point
    (:144), kappa is the wrapped heading change from the segment behind divided by
    ds; the segment "behind" point 0 wraps round to the last point (:135-137) and
    is then overwritten by kappa[1] (:152).
    """
    coords = np.asarray(coords, dtype=np.float64)
    n = coords.shape[0] - 1
    xy = coords[:, :2]
    here = xy[:-1]
    ahead = xy[1:] - here
    behind = here - np.concatenate([xy[-1:], xy[:-2]], axis=0)
    table = np.zeros((7, n))
    table[ROW_X] = here[:, 0]
    table[ROW_Y] = here[:, 1]
    table[ROW_WIDTH] = coords[1:, 2]
    table[ROW_PSI] = np.arctan2(ahead[:, 1], ahead[:, 0])
    table[ROW_DS] = np.sqrt(ahead[:, 0] ** 2 + ahead[:, 1] ** 2)
    heading_behind = np.arctan2(behind[:, 1], behind[:, 0])
    dpsi = wrap_to_pi(table[ROW_PSI] - heading_behind)
    kappa = dpsi / (table[ROW_DS] + EPS) + EPS
    kappa[0] = kappa[1]
    table[ROW_KAPPA] = kappa
    return table


def t2s(waypoint: Sequence[float], state: Sequence[float]) -> np.ndarray:
    """Cartesian -> Frenet w.r.t. one waypoint (dynamics.py:23-40)."""
    xr, yr, psir = waypoint
    x, y, psi = state
    e_y = math.cos(psir) * (y - yr) - math.sin(psir) * (x - xr)
    e_psi = wrap_to_pi(psi - psir)
    return np.array([e_y, float(e_psi), 0.0])


def s2t(table: np.ndarray, states: np.ndarray) -> np.ndarray:
    """Frenet -> Cartesian for n waypoints at once, returns 3 x n (dynamics.py:42-63)."""
    e_y = states[:, 0]
    psi = table[ROW_PSI]
    return np.stack(
        [table[ROW_X] - e_y * np.sin(psi), table[ROW_Y] + e_y * np.cos(psi), psi + states[:, 1]]
    )


def linearise(table: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Per-waypoint discrete spatial dynamics f[n,3], A[n,3,3], B[n,3,2] (dynamics.py:65-103)."""
    ds, kappa, v = table[ROW_DS], table[ROW_KAPPA], table[ROW_V]
    n = table.shape[1]
    A = np.tile(np.eye(3), (n, 1, 1))
    A[:, 0, 1] = ds
    A[:, 1, 0] = -(kappa**2) * ds
    A[:, 2, 0] = -kappa / (v * ds + EPS)
    B = np.zeros((n, 3, 2))
    B[:, 1, 1] = ds
    B[:, 2, 0] = -1.0 / (v**2 * ds + EPS)
    f = np.zeros((n, 3))
    f[:, 2] = 1.0 / (v * ds + EPS)
    return f, A, B


def vehicle_limits(wheelbase: float, width: float, delta_max: float, v_min: float, v_max: float):
    """Scalars of SpatialBicycleModel.__init__ (dynamics.py:10-21)."""
    k_max = np.tan(delta_max) / wheelbase
    return SimpleNamespace(
        length=wheelbase,
        width=width,
        delta_max=delta_max,
        margin=width / 2.0,
        min_u=np.array([v_min, -k_max]),
        max_u=np.array([v_max, k_max]),
    )


def control_qp(spatial_state: np.ndarray, table: np.ndarray, weights: Dict, limits) -> Dict:
    """Dense restatement of ControlSolver's QP (control/solvers/control.py:15-79,121-158).

    Decision vector z = [x_0 .. x_n (3 each) ; u_0 .. u_{n-1} (2 each)].
    Returns P's diagonal, q, dense A ((8n+6) x (5n+3)), l, u.
    """
    n = table.shape[1]
    nx, nu = 3, 2
    nz = nx * (n + 1) + nu * n
    Q = np.asarray(weights["step_cost"], dtype=np.float64)
    R = np.asarray(weights["r_term"], dtype=np.float64)
    QN = np.asarray(weights["final_cost"], dtype=np.float64)
    f, A_blk, B_blk = linearise(table)
    u_ref = np.stack([table[ROW_V], table[ROW_KAPPA]], axis=1)  # n x 2

    # equality rows: -x_0 = -x_init ; A_i x_i - x_{i+1} + B_i u_i = B_i u_ref_i - f_i   (:26-45)
    A_eq = np.zeros((nx * (n + 1), nz))
    A_eq[:, : nx * (n + 1)] = -np.eye(nx * (n + 1))
    for i in range(n):
        r = nx * (i + 1)
        A_eq[r : r + nx, nx * i : nx * (i + 1)] += A_blk[i]
        c = nx * (n + 1) + nu * i
        A_eq[r : r + nx, c : c + nu] = B_blk[i]
    uq = (np.einsum("nij,nj->ni", B_blk, u_ref) - f).ravel()
    l_eq = np.concatenate([-np.asarray(spatial_state, dtype=np.float64), uq])

    # box rows (:47-70,130-144)
    x_lo = np.tile([-np.inf, -np.inf, T_MIN], n + 1)
    x_hi = np.tile([np.inf, np.inf, np.inf], n + 1)
    x_lo[0] = spatial_state[0]
    x_hi[0] = spatial_state[0]
    half = table[ROW_WIDTH] / 2.0
    ey_lo = -half + limits.margin
    ey_hi = half - limits.margin
    x_lo[nx::nx] = ey_lo
    x_hi[nx::nx] = ey_hi
    x_ref = np.zeros(nx * (n + 1))
    x_ref[nx::nx] = (ey_lo + ey_hi) / 2.0
    u_lo = np.tile(limits.min_u, n)
    u_hi = np.tile(limits.max_u, n)
    u_lo[0::2] -= U_SLACK_V
    u_hi[0::2] += U_SLACK_V

    A = np.vstack([A_eq, np.eye(nz)])
    lower = np.concatenate([l_eq, x_lo, u_lo])
    upper = np.concatenate([l_eq, x_hi, u_hi])
    P_diag = np.concatenate([np.tile(Q, n), QN, np.tile(R, n)])
    q = -np.concatenate([np.tile(Q, n) * x_ref[:-nx], QN * x_ref[-nx:], np.tile(R, n) * u_ref.ravel()])
    return dict(P_diag=P_diag, q=q, A=A, l=lower, u=upper, x_ref=x_ref, u_ref=u_ref)


def speed_profile_qp(table: np.ndarray, constraints: Dict, end_velocity: Optional[float], localised: bool) -> Dict:
    """Inputs of the speed-profile QP (control/solvers/speed_profile.py:26-59,131-150).

    min 1/2 |v|^2 - v_hi' v   s.t.  a_min <= (v[i+1]-v[i])/(2 ds[i]) <= a_max,  v_min <= v <= v_hi
    """
    n = table.shape[1]
    kappa, ds = table[ROW_KAPPA], table[ROW_DS]
    v_min, v_max = constraints["v_min"], constraints["v_max"]
    if localised:
        v_hi = np.full(n, float(v_max))
    else:
        v_dyn = np.sqrt(constraints["ay_max"] / (np.abs(kappa) + EPS))
        v_dyn[np.abs(kappa) < constraints["ki_min"]] = v_max
        v_hi = np.maximum(v_min, np.minimum(v_dyn, v_max)) + 2.0
        if end_velocity is not None:
            v_hi[-1] = end_velocity
    D1 = np.zeros((n - 1, n))
    idx = np.arange(n - 1)
    D1[idx, idx] = -1.0 / (2.0 * ds[:-1])
    D1[idx, idx + 1] = 1.0 / (2.0 * ds[:-1])
    A = np.vstack([D1, np.eye(n)])
    lower = np.concatenate([np.full(n - 1, float(constraints["a_min"])), np.full(n, float(v_min))])
    upper = np.concatenate([np.full(n - 1, float(constraints["a_max"])), v_hi])
    return dict(P_diag=np.ones(n), q=-v_hi, A=A, l=lower, u=upper, v_hi=v_hi)


def speed_profile_exact(v_hi: np.ndarray, ds: np.ndarray, a_min: float, a_max: float, v_min: float) -> Optional[np.ndarray]:
    """The exact optimum of `speed_profile_qp` (control/solvers/speed_profile.py:26-59) without a solver - restatement of
    csrc/acmpc_admm.h exact_profile, float64 operation for operation.  The objective is 1/2 |v - v_hi|^2 + const and v_hi is
    the box's upper bound, so the optimum is the pointwise largest feasible profile: v_hi cut down by a forward pass
    (v[i+1] <= v[i] + 2 ds[i] a_max) and a backward pass (v[i] <= v[i+1] - 2 ds[i] a_min), each the prefix minimum of
    "ceiling + gaps in between" evaluated as a scan with doubling distances - the order of the roundings is part of the
    specification.  None where the passes do not apply (a_min > 0, a_max < 0, bad spacing, non-finite ceiling) or the
    problem is infeasible (some v below v_min)."""
    v_hi = np.asarray(v_hi, dtype=np.float64)
    ds = np.asarray(ds, dtype=np.float64)
    n = v_hi.shape[0]
    if not (a_max >= 0.0) or not (a_min <= 0.0):
        return None
    with np.errstate(invalid="ignore", over="ignore"):
        v = v_hi.copy()
        gap = np.zeros(n)
        gap[1:] = (2.0 * ds[:n - 1]) * a_max
        d = 1
        while d < n:
            cand = v[:-d] + gap[d:]
            new_v, new_gap = v.copy(), gap.copy()
            new_v[d:] = np.where(cand < v[d:], cand, v[d:])
            new_gap[d:] = gap[d:] + gap[:-d]
            v, gap = new_v, new_gap
            d *= 2
        gap = np.zeros(n)
        gap[:n - 1] = (-2.0 * ds[:n - 1]) * a_min
        d = 1
        while d < n:
            cand = v[d:] + gap[:-d]
            new_v, new_gap = v.copy(), gap.copy()
            new_v[:-d] = np.where(cand < v[:-d], cand, v[:-d])
            new_gap[:-d] = gap[:-d] + gap[d:]
            v, gap = new_v, new_gap
            d *= 2
        if not (np.all(v >= v_min) and np.all(v <= v_hi)) or not np.all((ds[:n - 1] > 0.0) & np.isfinite(ds[:n - 1])):
            return None
    return v


def osqp_restated(P_diag, q, A, l, u, max_iter=4000, rho=0.1, sigma=1e-6, alpha=1.6,
                  eps_abs=1e-3, eps_rel=1e-3, check_every=25, adaptive_rho=False) -> SimpleNamespace:
    """ADMM of Stellato et al., "OSQP: an operator splitting solver for quadratic programs"
    (Math. Prog. Comp. 2020), Algorithm 1 - dense, no scaling, no rho adaptation, equality rows
    get 1e3*rho as in the paper's section 5.2.  `osqp` is a third-party dependency of the reference
    with no pinned version (requirements.txt:2) and is absent here, so this restates the published
    algorithm; solutions are parity-UNPINNED and only used as a sanity oracle (is the sampled
    optimum close to the QP optimum?).  Termination tolerances are OSQP's documented defaults.
    `adaptive_rho`: the paper's section 5.2 step-size update (rho scaled by the square root of the ratio of the
    normalised primal and dual residuals when it is off by more than a factor 5, the KKT matrix refactored) at the
    residual checks - what lets the tight-tolerance sanity solves converge on the nearly straight scenarios.
    """
    P_diag = np.asarray(P_diag, dtype=np.float64)
    m, nvar = A.shape
    from scipy.linalg import cho_factor, cho_solve

    def factor(rho_now):
        vec = np.where(l == u, 1e3 * rho_now, rho_now)
        return vec, cho_factor(np.diag(P_diag + sigma) + A.T @ (vec[:, None] * A), lower=True)

    rho_vec, chol = factor(rho)

    def kkt_solve(rhs):   # two triangular solves (a general solve of the factor would refactor it every iteration)
        return cho_solve(chol, rhs)

    x = np.zeros(nvar)
    z = np.zeros(m)
    y = np.zeros(m)
    status = "maximum iterations reached"
    it = 0
    for it in range(1, max_iter + 1):
        x_t = kkt_solve(sigma * x - q + A.T @ (rho_vec * z - y))
        z_t = A @ x_t
        x = alpha * x_t + (1 - alpha) * x
        z_relaxed = alpha * z_t + (1 - alpha) * z
        z_new = np.clip(z_relaxed + y / rho_vec, l, u)
        y = y + rho_vec * (z_relaxed - z_new)
        z = z_new
        if it % check_every == 0:
            Ax = A @ x
            r_prim = np.max(np.abs(Ax - z))
            r_dual = np.max(np.abs(P_diag * x + q + A.T @ y))
            e_prim = eps_abs + eps_rel * max(np.max(np.abs(Ax)), np.max(np.abs(z)))
            e_dual = eps_abs + eps_rel * max(np.max(np.abs(P_diag * x)), np.max(np.abs(A.T @ y)), np.max(np.abs(q)))
            if r_prim <= e_prim and r_dual <= e_dual:
                status = "solved"
                break
            if adaptive_rho and it % (4 * check_every) == 0:
                n_prim = r_prim / max(np.max(np.abs(Ax)), np.max(np.abs(z)), 1e-12)
                n_dual = r_dual / max(np.max(np.abs(P_diag * x)), np.max(np.abs(A.T @ y)), np.max(np.abs(q)), 1e-12)
                scale = np.sqrt(n_prim / max(n_dual, 1e-300))
                if scale > 5.0 or scale < 0.2:
                    rho = float(np.clip(rho * scale, 1e-6, 1e6))
                    rho_vec, chol = factor(rho)
    return SimpleNamespace(x=x, y=y, info=SimpleNamespace(status=status, iter=it))


def kinematic_x_dot(delta, states: np.ndarray, velocity, wheel_base: float) -> np.ndarray:
    """Rear-axle kinematic bicycle derivative, batched over particles (localiser.py:77-95)."""
    phi = states[:, 2]
    out = np.zeros_like(states)
    out[:, 0] = velocity * np.cos(phi)
    out[:, 1] = velocity * np.sin(phi)
    out[:, 2] = velocity * np.tan(delta) / wheel_base
    return out


def nearest_waypoint(points: np.ndarray, track: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Euclidean nearest neighbour, first minimum - what KDTree.query returns (localiser.py:282-289)."""
    d2 = ((points[:, None, :] - track[None, :, :]) ** 2).sum(axis=2)
    idx = np.argmin(d2, axis=1)
    return np.sqrt(d2[np.arange(len(points)), idx]), idx


def heading_offset(track: np.ndarray, idx: np.ndarray, headings: np.ndarray) -> np.ndarray:
    """|wrap(track heading at idx - particle heading)|, index mod (len-1) (localiser.py:291-318)."""
    m = len(track) - 1
    here = track[np.mod(idx, m)]
    nxt = track[np.mod(idx + 1, m)]
    heading = np.arctan2(nxt[:, 1] - here[:, 1], nxt[:, 0] - here[:, 0])
    return np.abs((heading - headings + np.pi) % (2 * np.pi) - np.pi)


def estimate_location(scores: np.ndarray, states: np.ndarray) -> np.ndarray:
    """Score-weighted mean with the NaN -> uniform fallback (localiser.py:572-579)."""
    w = scores.reshape(-1, 1)
    est = (states[:, :3] * w).sum(axis=0) / w.sum()
    if np.any(np.isnan(est)):
        w = np.full_like(w, 1.0 / w.shape[0])
        est = (states[:, :3] * w).sum(axis=0) / w.sum()
    return est


def select_command(cum_time: np.ndarray, commands: np.ndarray, elapsed: float) -> np.ndarray:
    """TemporalCommandSelector.get_command (commands.py:20-38) incl. the -1 -> last-row wrap."""
    d = cum_time - elapsed
    i = int(np.argmin(np.abs(d)))
    if d[i] > 0:
        i -= 1
    if i >= len(commands):
        i = len(commands) - 1
    return commands[i]


def interpolate_command(cum_time: np.ndarray, commands: np.ndarray, elapsed: float) -> np.ndarray:
    """TemporalCommandInterpolator.get_command (commands.py:54-99); `commands` is n x 2."""
    d = cum_time - elapsed
    a = int(np.argmin(np.abs(d)))
    if a == 0 or a == len(commands) - 1:
        return commands[a]
    b = a + 1 if d[a] < 0 else a - 1
    ta, tb = cum_time[a], cum_time[b]
    return commands[a] * ((tb - elapsed) / (tb - ta)) + commands[b] * ((elapsed - ta) / (tb - ta))


def closest_command_index(cum_time: np.ndarray, elapsed: float) -> Tuple[int, float]:
    """TemporalCommandInterpolator._get_closet_command_index (commands.py:71-74)."""
    d = cum_time - elapsed
    i = int(np.argmin(np.abs(d)))
    return i, float(d[i])


def downsample_centreline(centreline: np.ndarray, horizon: int) -> np.ndarray:
    """500 x 2 perception centreline -> H x 3 reference path (controller.py:256-267)."""
    step = int(len(centreline) / horizon)
    pts = centreline[0::step]
    return np.stack([pts[:, 0], pts[:, 1], np.linspace(10.0, 6.0, horizon)]).T


# ---------------------------------------------------------------------------
# Build-defined sampling composition ("mode S" / "mode T")
# ---------------------------------------------------------------------------
def input_box(limits) -> Tuple[np.ndarray, np.ndarray]:
    """u-box of the QP: [min_u - (0.1, 0), max_u + (0.1, 0)] (control.py:130-139)."""
    lo = np.array([limits.min_u[0] - U_SLACK_V, limits.min_u[1]])
    hi = np.array([limits.max_u[0] + U_SLACK_V, limits.max_u[1]])
    return lo, hi


def coefficients_spatial(table: np.ndarray, margin: float) -> np.ndarray:
    """Mode-S per-step record [ds, a21, a31, b31, f3, v_ref, k_ref, ey_lo, ey_hi, 0,0,0].

    Entries are the non-trivial elements of linearise()'s A, B, f (dynamics.py:65-103) and the
    corridor bounds of x_{i+1} (control.py:57-60), computed in float64, returned as float32.
    """
    f, A, B = linearise(table)
    n = table.shape[1]
    c = np.zeros((n, COEF_STRIDE_S))
    c[:, CS_DS] = A[:, 0, 1]
    c[:, CS_A21] = A[:, 1, 0]
    c[:, CS_A31] = A[:, 2, 0]
    c[:, CS_B31] = B[:, 2, 0]
    c[:, CS_F3] = f[:, 2]
    c[:, CS_VREF] = table[ROW_V]
    c[:, CS_KREF] = table[ROW_KAPPA]
    c[:, CS_EYLO] = -table[ROW_WIDTH] / 2.0 + margin
    c[:, CS_EYHI] = table[ROW_WIDTH] / 2.0 - margin
    return c.astype(np.float32)


def coefficients_temporal(table: np.ndarray, margin: float) -> np.ndarray:
    """Mode-T waypoint record [x, y, cos psi, sin psi, psi, k_ref, v_ref, w/2 - margin] as float32."""
    n = table.shape[1]
    c = np.zeros((n, COEF_STRIDE_T))
    c[:, CT_X] = table[ROW_X]
    c[:, CT_Y] = table[ROW_Y]
    c[:, CT_COS] = np.cos(table[ROW_PSI])
    c[:, CT_SIN] = np.sin(table[ROW_PSI])
    c[:, CT_PSI] = table[ROW_PSI]
    c[:, CT_KREF] = table[ROW_KAPPA]
    c[:, CT_VREF] = table[ROW_V]
    c[:, CT_HALF] = table[ROW_WIDTH] / 2.0 - margin
    return c.astype(np.float32)


def _quad(w, a):
    # (w*a)*a  - fixed association, no FMA
    return (w * a) * a


def fma32(a, b, c):
    """IEEE-754 fmaf on float32 arrays: round_to_float32(a * b + c) with ONE rounding, bit for bit.

    The product of two float32 values is exact in float64 (48 <= 53 significant bits).  The float64 sum p + c is
    rounded; TwoSum (Knuth) recovers its rounding error exactly, and when that error is not zero the sum is moved to
    the neighbouring float64 whose last mantissa bit is odd ("round to odd", Boldo & Melquiond 2008): rounding THAT
    to float32 equals rounding the exact a * b + c to float32, because float64 carries more than 24 + 2 bits.
    Infinities and NaNs pass through the float64 operations unchanged."""
    a64 = np.asarray(a, dtype=np.float32).astype(np.float64)
    b64 = np.asarray(b, dtype=np.float32).astype(np.float64)
    c64 = np.asarray(c, dtype=np.float32).astype(np.float64)
    with np.errstate(all="ignore"):
        p = a64 * b64
        raw = p + c64
        s = np.atleast_1d(raw)
        bb = s - p
        err = (p - (s - bb)) + (c64 - bb)
        inexact = np.isfinite(s) & (err != 0.0) & ((s.view(np.int64) & 1) == 0)
        odd = np.nextafter(s, np.where(err > 0.0, np.inf, -np.inf))
        out = np.where(inexact, odd, s).astype(np.float32)
    return out.reshape(np.shape(raw)) if np.ndim(raw) else np.float32(out[0])


def _fma_for(dtype):
    """The fused multiply-add of the mode-T specification in `dtype`: exact fmaf in float32; in float64 (drift
    reports only) a plain multiply and add."""
    if dtype is np.float32:
        return fma32
    return lambda a, b, c: a * b + c


def _hinge2(lo_minus_x, x_minus_hi, zero):
    # distance outside [lo, hi]: at most one side of a non-degenerate interval can be violated.  IEEE maxNum (a NaN
    # operand is dropped) - C's fmaxf and the GPU's v_max_f32; np.maximum would propagate the NaN instead
    v = np.fmax(np.fmax(lo_minus_x, x_minus_hi), zero)
    return v * v


def rollout_spatial(x0, coef, U, Q, R, QN, u_lo, u_hi, w_bound, dtype=np.float32, return_states=False):
    """Mode S: x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i with the a9 cost and bounds.

    x0[3], coef[n,12] (coefficients_spatial), U[N,n,2] = (v, kappa) -> cost[N], viol[N] (0 = feasible)
    and optionally X[N,n+1,3].  All arithmetic in `dtype`, candidates vectorised, steps sequential.
    """
    T = dtype
    U = np.asarray(U, dtype=T)
    coef = np.asarray(coef, dtype=T)
    N, n, _ = U.shape
    Q, R, QN = (np.asarray(a, dtype=T) for a in (Q, R, QN))
    u_lo, u_hi = np.asarray(u_lo, dtype=T), np.asarray(u_hi, dtype=T)
    half, zero, tmin, wb = T(0.5), T(0.0), T(T_MIN), T(w_bound)
    ey = np.full(N, T(x0[0]), dtype=T)
    ep = np.full(N, T(x0[1]), dtype=T)
    t = np.full(N, T(x0[2]), dtype=T)
    J = np.zeros(N, dtype=T)
    V = np.zeros(N, dtype=T)
    X = np.zeros((N, n + 1, 3), dtype=T) if return_states else None
    for i in range(n):
        c = coef[i]
        v, k = U[:, i, 0], U[:, i, 1]
        if return_states:
            X[:, i, 0], X[:, i, 1], X[:, i, 2] = ey, ep, t
        dv = v - c[CS_VREF]
        dk = k - c[CS_KREF]
        s = _quad(Q[0], ey)
        s = s + _quad(Q[1], ep)
        s = s + _quad(Q[2], t)
        r = _quad(R[0], dv)
        r = r + _quad(R[1], dk)
        J = J + half * (s + r)
        V = V + _hinge2(u_lo[0] - v, v - u_hi[0], zero)
        V = V + _hinge2(u_lo[1] - k, k - u_hi[1], zero)
        ey_n = ey + c[CS_DS] * ep
        ep_n = (ep + c[CS_A21] * ey) + c[CS_DS] * dk
        t_n = ((t + c[CS_A31] * ey) + c[CS_B31] * dv) + c[CS_F3]
        ey, ep, t = ey_n, ep_n, t_n
        V = V + _hinge2(c[CS_EYLO] - ey, ey - c[CS_EYHI], zero)
        tv = np.fmax(tmin - t, zero)
        V = V + tv * tv
    if return_states:
        X[:, n, 0], X[:, n, 1], X[:, n, 2] = ey, ep, t
    s = _quad(QN[0], ey)
    s = s + _quad(QN[1], ep)
    s = s + _quad(QN[2], t)
    J = J + half * s
    cost = J + wb * V
    return (cost, V, X) if return_states else (cost, V)


# --- spec trigonometry (Cody-Waite reduction by pi + odd / even polynomials on [-pi/2, pi/2] fitted for this build) ---
INV_PI = 0.3183098861837907
PI_HI = 3.140625  # 9 significant bits: k*PI_HI is exact for |k| < 2^15
PI_LO = 9.67653589793e-4  # pi - PI_HI
# sin r = r + r^3 (S0 + S1 r^2 + S2 r^4 + S3 r^6),  cos r = 1 + r^2 (C0 + C1 r^2 + ... + C4 r^8): Lawson-weighted least
# squares on [-pi/2, pi/2], coefficients rounded to float32; |error| < 1.5e-7 evaluated in float32 with the FMAs below
SIN_C = (-0.16666656732559204, 0.008333016186952591, -0.00019806546333711594, 2.59990065387683e-06)
COS_C = (-0.5, 0.04166664183139801, -0.0013888402609154582, 2.4761806344031356e-05, -2.607563374112942e-07)
PI_F = 3.14159265358979
TWO_PI_F = 6.28318530717959
INV_TWO_PI_F = 0.159154943091895


ROUND_MAGIC = 12582912.0  # 1.5 * 2^23: adding it to |y| < 2^22 rounds y to the nearest-even integer in the mantissa


def sincos_spec(phi, dtype=np.float32):
    """sin/cos with a fixed instruction sequence (bit-identical on CPU and GPU in float32): phi = k pi + r with
    k = rint(phi / pi), sin phi = (-1)^k sin r, cos phi = (-1)^k cos r, every multiply-add fused."""
    T = dtype
    fma = _fma_for(T)
    phi = np.asarray(phi, dtype=T)
    if T is np.float32:
        # round-to-integer WITHOUT a float -> int conversion (whose out-of-range result differs between x86, NumPy
        # and the GPU): t = fma(phi, 1/pi, 1.5 * 2^23) holds rint(y) in its low mantissa bits, k = t - 1.5 * 2^23.
        # Identical to rint for |phi| < 1.3e7 rad; beyond that still one fixed bit pattern everywhere.
        t = np.atleast_1d(fma(phi, T(INV_PI), T(ROUND_MAGIC)))
        odd = ((t.view(np.int32) & 1) != 0).reshape(np.shape(phi))
        k = t.reshape(np.shape(phi)) - T(ROUND_MAGIC)
    else:  # float64 drift reports only
        k = np.rint(phi * T(INV_PI))
        odd = (np.where(np.isfinite(k), k, 0.0).astype(np.int64) & 1) != 0
    r = fma(-k, T(PI_HI), phi)
    r = fma(-k, T(PI_LO), r)
    r2 = r * r
    ps = fma(r2, T(SIN_C[3]), T(SIN_C[2]))
    ps = fma(r2, ps, T(SIN_C[1]))
    ps = fma(r2, ps, T(SIN_C[0]))
    s = fma(r * r2, ps, r)
    pc = fma(r2, T(COS_C[4]), T(COS_C[3]))
    pc = fma(r2, pc, T(COS_C[2]))
    pc = fma(r2, pc, T(COS_C[1]))
    pc = fma(r2, pc, T(COS_C[0]))
    c = fma(r2, pc, T(1.0))
    sin = np.where(odd, -s, s)
    cos = np.where(odd, -c, c)
    return sin.astype(T), cos.astype(T)


def wrap_spec(angle, dtype=np.float32):
    """Angle difference into [-pi, pi]: a - 2 pi rint(a / 2 pi), the integer read off the magic-number sum
    fma(a, 1 / 2 pi, 1.5 * 2^23) (csrc/acmpc_device.h: wrap_spec; no float -> int conversion, no division)."""
    T = dtype
    fma = _fma_for(T)
    magic = T(12582912.0) if T == np.float32 else T(6755399441055744.0)   # 1.5 * 2^23 / 1.5 * 2^52
    a = np.asarray(angle, dtype=T)
    q = fma(a, T(INV_TWO_PI_F), magic) - magic
    return fma(-q, T(TWO_PI_F), a)


def rollout_temporal(pose0, wp, U, Q, R, QN, u_lo, u_hi, w_bound, dt, dtype=np.float32,
                     return_states=False, libm_trig=False, nn_window=None):
    """Mode T: Cartesian kinematic Euler rollout (localiser.py:66-95 with phi_dot = v*kappa, kappa =
    tan(delta)/L), nearest-waypoint projection (first minimum of the squared distance, localiser.py:282-289 - compared
    through the search key e_m = |p - w_m|^2 - |p|^2 evaluated as two fused multiply-adds, see below),
    Frenet errors (dynamics.py:23-40), a9 cost weights and bounds.

    pose0 = (X, Y, phi), wp[n,8] (coefficients_temporal), U[N,n,2] = (v, kappa).
    nn_window = (back, ahead): search only the W = back + ahead + 1 consecutive waypoints starting at
    clamp(j_prev - back, 0, n - W) (j_prev = previous step's nearest index, 0 at the start) instead of all n - a
    build-defined shortcut, equal to the exhaustive search whenever progress along the path is slower than the
    window (checked in tests).
    """
    T = dtype
    U = np.asarray(U, dtype=T)
    wp = np.asarray(wp, dtype=T)
    N, n, _ = U.shape
    Q, R, QN = (np.asarray(a, dtype=T) for a in (Q, R, QN))
    u_lo, u_hi = np.asarray(u_lo, dtype=T), np.asarray(u_hi, dtype=T)
    half, zero, wb, dtT = T(0.5), T(0.0), T(w_bound), T(dt)
    fma = _fma_for(T)
    hQ, hR, hQN = half * Q, half * R, half * QN   # the halved weights (products in `dtype`, exact)
    # the path's own frame (csrc/acmpc_device.h: start_temporal): every position relative to waypoint 0, float32 differences
    # - the search key cancels catastrophically far from the origin; poses are reported back in the caller's frame
    ox, oy = wp[0, CT_X], wp[0, CT_Y]
    wx, wy = wp[:, CT_X] - ox, wp[:, CT_Y] - oy
    X = np.full(N, T(pose0[0]) - ox, dtype=T)
    Y = np.full(N, T(pose0[1]) - oy, dtype=T)
    phi = np.full(N, T(pose0[2]), dtype=T)
    J = np.zeros(N, dtype=T)
    V = np.zeros(N, dtype=T)
    S = np.zeros((N, n + 1, 3), dtype=T) if return_states else None
    J_idx = np.zeros((N, n), dtype=np.int32) if return_states else None
    ey = np.zeros(N, dtype=T)
    ep = np.zeros(N, dtype=T)
    j_prev = np.zeros(N, dtype=np.int64)
    if return_states:
        S[:, 0, 0], S[:, 0, 1], S[:, 0, 2] = X + ox, Y + oy, phi
    # the nearest-waypoint search key (csrc/acmpc_device.h: search_key): e_m = fma(Y, b_m, fma(X, a_m, c_m)) with
    # a = -2 x, b = -2 y, c = fma(y, y, x x) - the squared distance less |p|^2, which no waypoint's share of changes
    key_a, key_b = T(-2.0) * wx, T(-2.0) * wy
    key_c = fma(wy, wy, wx * wx)
    # the waypoint rows as the kernels derive them once per workgroup (stage_temporal_tables): e_y = c (Y - y) - s (X - x)
    # becomes fma(c, Y, fma(-s, X, s x - c y))
    row_k = fma(wp[:, CT_SIN], wx, -(wp[:, CT_COS] * wy))
    row_ns = -wp[:, CT_SIN]
    S0, S1, S2, S3 = (np.zeros(N, dtype=T) for _ in range(4))   # sums of e_y^2, e_psi^2, dv^2, dkappa^2
    for i in range(n):
        v, k = U[:, i, 0], U[:, i, 1]
        if libm_trig:
            sn, cs = np.sin(phi).astype(T), np.cos(phi).astype(T)
        else:
            sn, cs = sincos_spec(phi, T)
        Xn = fma(v * cs, dtT, X)
        Yn = fma(v * sn, dtT, Y)
        phin = fma(v * k, dtT, phi)
        X, Y, phi = Xn, Yn, phin
        best = np.full(N, np.inf, dtype=T)
        if nn_window is None:
            j = np.zeros(N, dtype=np.int64)
            for w in range(n):
                d = fma(Y, key_b[w], fma(X, key_a[w], key_c[w]))
                better = d < best
                best = np.where(better, d, best)
                j = np.where(better, w, j)
        else:
            back, ahead = nn_window
            width = back + ahead + 1
            lo = np.maximum(np.minimum(j_prev - back, n - width), 0)
            hi = np.minimum(lo + width, n) - 1
            j = lo.copy()
            for m in range(width):
                w = np.minimum(lo + m, hi)
                d = fma(Y, key_b[w], fma(X, key_a[w], key_c[w]))
                better = d < best
                best = np.where(better, d, best)
                j = np.where(better, w, j)
        j_prev = j
        g = wp[j]
        ey = fma(g[:, CT_COS], Y, fma(row_ns[j], X, row_k[j]))
        if libm_trig:
            ep = wrap_to_pi(phi - g[:, CT_PSI]).astype(T)
        else:
            ep = wrap_spec(phi - g[:, CT_PSI], T)
        dv = v - g[:, CT_VREF]
        dk = k - g[:, CT_KREF]
        # the stage cost's squares; its weights are applied once, after the horizon
        S0 = fma(ey, ey, S0)
        S1 = fma(ep, ep, S1)
        S2 = fma(dv, dv, S2)
        S3 = fma(dk, dk, S3)
        # excess over the input box: x - med3(x, lo, hi) (v_med3_f32: a NaN x gives min(lo, hi))
        hv = v - np.fmin(np.fmax(v, u_lo[0]), u_hi[0])
        V = fma(hv, hv, V)
        hk = k - np.fmin(np.fmax(k, u_lo[1]), u_hi[1])
        V = fma(hk, hk, V)
        hc = np.fmax(np.abs(ey) - g[:, CT_HALF], zero)           # outside the corridor |e_y| <= w/2 - margin
        V = fma(hc, hc, V)
        if return_states:
            S[:, i + 1, 0], S[:, i + 1, 1], S[:, i + 1, 2] = X + ox, Y + oy, phi
            J_idx[:, i] = j
    tN = T(n) * dtT
    J = hQ[0] * S0
    J = fma(hQ[1], S1, J)
    J = fma(hR[0], S2, J)
    J = fma(hR[1], S3, J)
    s = (hQN[0] * ey) * ey
    s = fma(hQN[1] * ep, ep, s)
    s = fma(hQN[2] * tN, tN, s)
    J = J + s
    cost = fma(wb, V, J)
    return (cost, V, S, J_idx) if return_states else (cost, V)


def pick_best(costs: np.ndarray) -> Tuple[int, float]:
    """argmin with lowest index on ties; non-finite costs rank as +inf (never selected unless all are)."""
    c = np.where(np.isfinite(costs), costs, np.inf)
    i = int(np.argmin(c))
    return i, float(costs[i])


def softmin_weights(costs: np.ndarray, lam: float, dtype=np.float32) -> np.ndarray:
    """exp(-(c - c_min)/lambda); non-finite costs get weight 0."""
    T = dtype
    c = np.where(np.isfinite(costs), costs, np.inf).astype(T)
    return np.exp(-(c - c.min()) / T(lam)).astype(T)


def softmin_mean(costs: np.ndarray, U: np.ndarray, lam: float) -> np.ndarray:
    """Score-weighted mean of the control sequences in a16's form sum(w*u)/sum(w) (localiser.py:572-579),
    accumulated in float64 so the result does not depend on summation order."""
    w = softmin_weights(costs, lam).astype(np.float64)
    return np.tensordot(w, U.astype(np.float64), axes=(0, 0)) / w.sum()


def qp_objective(P_diag, q, z):
    return 0.5 * float(np.dot(P_diag * z, z)) + float(np.dot(q, z))


def pack_decision_vector(X: np.ndarray, U: np.ndarray) -> np.ndarray:
    """[x_0..x_n ; u_0..u_{n-1}] layout of dec.x (control.py:121-158, spatial_mpc.py:193-202)."""
    return np.concatenate([X.ravel(), U.ravel()])


# ---------------------------------------------------------------------------
# On-device candidate generation (build-defined)
# ---------------------------------------------------------------------------
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
SAMPLE_KNOTS = 8


def philox4x32_10(counter: np.ndarray, key: np.ndarray) -> np.ndarray:
    """Philox4x32-10 of Salmon, Moraes, Dror & Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11),
    vectorised: counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32.  Integer arithmetic: exact."""
    c = np.asarray(counter, dtype=np.uint64) & 0xFFFFFFFF
    k = np.asarray(key, dtype=np.uint64) & 0xFFFFFFFF
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(PHILOX_W1)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def sample_segments(n: int) -> np.ndarray:
    """Per step: (left knot, weight of the left knot); the next knot gets 1 - weight (raised cosine)."""
    width = (n - 1) / (SAMPLE_KNOTS - 1)
    pos = np.arange(n) / width
    k0 = np.minimum(np.floor(pos), SAMPLE_KNOTS - 2)
    w0 = 0.5 * (1.0 + np.cos(np.pi * (pos - k0)))
    return np.stack([k0, w0], axis=1).astype(np.float32)


LOG_C = (7.0376836292e-2, -1.1514610310e-1, 1.1676998740e-1, -1.2420140846e-1, 1.4249322787e-1, -1.6668057665e-1,
         2.0000714765e-1, -2.4999993993e-1, 3.3333331174e-1)   # Cephes logf (Moshier): ln(1 + f) = f - f^2/2 + f^3 P(f)
LN2_HI, LN2_LO = 0.693359375, -2.12194440e-4


def log_spec(u):
    """ln u for float32 u in (0, 1] with the fixed operation sequence of csrc/acmpc_device.h: log_spec - u = m 2^e with
    m in [sqrt(1/2), sqrt(2)) read off the bit pattern, f = m - 1, every multiply-add one fmaf.  Bit-identical to the kernel."""
    T = np.float32
    u = np.atleast_1d(np.asarray(u, dtype=T))
    bits = u.view(np.int32)
    e = (bits - np.int32(0x3F3504F3)) >> 23                      # arithmetic shift
    m = (bits - (e << 23)).astype(np.int32).view(T)
    f = m - T(1.0)
    ef = e.astype(T)
    z = f * f
    p = fma32(T(LOG_C[0]), f, T(LOG_C[1]))
    for c in LOG_C[2:]:
        p = fma32(p, f, T(c))
    y = (f * z) * p
    y = fma32(ef, T(LN2_LO), y)
    y = fma32(T(-0.5), z, y)
    return fma32(ef, T(LN2_HI), f + y)


def box_muller_spec(u1, u2):
    """Two standard normals from two uniforms, bit-identical to csrc/acmpc_device.h: box_muller (round 4): spec'd
    logarithm, correctly rounded square root, sin / cos of 2 pi u2 reduced in turns with the rollout's polynomials."""
    T = np.float32
    u1, u2 = np.atleast_1d(np.asarray(u1, dtype=T)), np.atleast_1d(np.asarray(u2, dtype=T))
    with np.errstate(all="ignore"):
        radius = np.sqrt(T(-2.0) * log_spec(u1)).astype(T)
    t = fma32(u2, T(2.0), T(ROUND_MAGIC))
    k = t - T(ROUND_MAGIC)
    r = fma32(k, T(-0.5), u2)
    x = r * T(TWO_PI_F)
    x2 = x * x
    ps = fma32(x2, T(SIN_C[3]), T(SIN_C[2]))
    ps = fma32(x2, ps, T(SIN_C[1]))
    ps = fma32(x2, ps, T(SIN_C[0]))
    sn = fma32(x * x2, ps, x)
    pc = fma32(x2, T(COS_C[4]), T(COS_C[3]))
    pc = fma32(x2, pc, T(COS_C[2]))
    pc = fma32(x2, pc, T(COS_C[1]))
    pc = fma32(x2, pc, T(COS_C[0]))
    cs = fma32(x2, pc, T(1.0))
    odd = (np.atleast_1d(t).view(np.int32) & 1) != 0
    cs, sn = np.where(odd, -cs, cs).astype(T), np.where(odd, -sn, sn).astype(T)
    return radius * cs, radius * sn


def uniform_open(bits):
    """32 random bits -> float32 in (0, 1]: (bits >> 8) 2^-24 + 2^-25 in float32 (csrc/acmpc_device.h: uniform_open)."""
    return (np.asarray(bits, dtype=np.uint32) >> 8).astype(np.float32) * np.float32(2.0**-24) + np.float32(2.0**-25)


def candidate_normals(n_candidates, index_offset, problem, round_, seed):
    """The SAMPLE_KNOTS x 2 standard normals of every candidate: Philox4x32-10 at counter (global index, problem, round,
    draw q), two Box-Muller pairs per draw (csrc/acmpc_device.h: draw_normals).  float32, exact."""
    gidx = (np.arange(n_candidates, dtype=np.uint64) + np.uint64(index_offset)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    z = np.zeros((n_candidates, SAMPLE_KNOTS, 2), dtype=np.float32)
    for q in range(SAMPLE_KNOTS // 2):
        ctr = np.stack([gidx, np.full_like(gidx, problem), np.full_like(gidx, round_), np.full_like(gidx, q)], axis=1)
        r = philox4x32_10(ctr, np.broadcast_to(key, (n_candidates, 2)))
        u = uniform_open(r)
        for half in range(2):
            z[:, 2 * q + half, 0], z[:, 2 * q + half, 1] = box_muller_spec(u[:, 2 * half], u[:, 2 * half + 1])
    return gidx, z


def sample_candidates(centre, u_ref, n_candidates, index_offset, problem, round_, seed, sigma, u_lo, u_hi, u_extra=None):
    """Restates csrc sample_kernel / the fused rounds' candidate generation for ONE problem, bit for bit (round 4: the
    device's normals are specified - box_muller_spec - and the blend is float32 with one fixed association, no FMA):
        U_c[i] = clip(centre[i] + (sigma amp_c) (w0_i z_c[k0_i] + (1 - w0_i) z_c[k0_i + 1]))
    amp_c = ((c & 7) + 1) / 8; candidate 0 = the centre, candidate 1 = `u_ref` (when given), candidate 2 = `u_extra` (when
    given: the LQ plan, lq_plan()) - each passed through the same clip.  centre / u_ref / u_extra [n, 2] -> U [N, n, 2] float32."""
    T = np.float32
    centre = np.asarray(centre, dtype=T)
    n = centre.shape[0]
    gidx, z = candidate_normals(n_candidates, index_offset, problem, round_, seed)
    seg = sample_segments(n)
    k0 = seg[:, 0].astype(int)
    w0 = seg[:, 1].astype(T)
    w1 = T(1.0) - w0
    amp = ((gidx & 7) + 1).astype(T) * T(0.125)
    amp[gidx == 0] = T(0.0)
    base = np.broadcast_to(centre, (n_candidates, n, 2)).copy()
    if u_ref is not None:
        base[gidx == 1] = np.asarray(u_ref, dtype=T)
        amp[gidx == 1] = T(0.0)
    if u_extra is not None:
        base[gidx == 2] = np.asarray(u_extra, dtype=T)
        amp[gidx == 2] = T(0.0)
    sig = np.asarray(sigma, dtype=T)
    lo, hi = np.asarray(u_lo, dtype=T), np.asarray(u_hi, dtype=T)
    # blend_control: v = cv + (sigma_v amp) (w0 z0v + w1 z1v), then fmin(fmax(v, lo), hi)
    noise = w0[None, :, None] * z[:, k0, :] + w1[None, :, None] * z[:, k0 + 1, :]          # [N, n, 2]
    scale = sig[None, :] * amp[:, None]                                                    # [N, 2]
    U = base + scale[:, None, :] * noise
    return np.fmin(np.fmax(U, lo), hi).astype(T)


def optimize_restated(mode, start, coef, centre, u_ref, n_candidates, rounds, sigma, shrink, seed, Q, R, QN, u_lo, u_hi,
                      w_bound=1.0e6, dt=0.05, nn_window=None, extra=None, problem=0):
    """One whole solve of the sampling controller, restated (round 4): what acmpc_optimize and the rounds of
    acmpc_control_tick compute for ONE problem, bit for bit -
        for r in 0 .. rounds - 1:   candidates = sample_candidates(centre_r, u_ref, sigma shrink^r, round r;
                                                 candidate 2 = `extra` - the LQ plan - in the LAST round)
                                    costs      = rollout_spatial / rollout_temporal(candidates)        (float32 spec order)
                                    winner     = first minimum, non-finite costs last (pick_best)
                                    centre_r+1 = the winner's controls
    - the seam the tick replaces in the reference is SpatialMPC.get_control's solver call (spatial_mpc.py:185-217).
    `start`: mode 0 the Frenet state, mode 1 the pose; `coef`: the packed float32 table ([n, 12] / [n, 8]);
    centre, u_ref, extra: [n, 2].  Returns the winner of the last round as the record holds it:
    dict(cost, violation, n_feasible, index, u [n, 2], x [n + 1, 3]) plus `winners`, the index per round."""
    T = np.float32
    centre = np.asarray(centre, dtype=T)
    winners = []
    scale = 1.0
    out = None
    for r in range(rounds):
        last = r + 1 == rounds
        spread = (T(float(sigma[0]) * scale), T(float(sigma[1]) * scale))      # (double product, one rounding: make_spec)
        U = sample_candidates(centre, u_ref, n_candidates, 0, problem, r, seed, spread, u_lo, u_hi,
                              u_extra=extra if last else None)
        if mode == 0:
            cost, viol, X = rollout_spatial(start, coef, U, Q, R, QN, u_lo, u_hi, w_bound, dtype=T, return_states=True)
        else:
            cost, viol, X, _ = rollout_temporal(start, coef, U, Q, R, QN, u_lo, u_hi, w_bound, dt, dtype=T,
                                                return_states=True, nn_window=nn_window)
        best, _ = pick_best(cost)
        winners.append(best)
        centre = U[best]
        out = dict(cost=cost[best], violation=viol[best], n_feasible=int(np.count_nonzero(viol == 0)), index=best,
                   u=U[best].copy(), x=X[best].copy())
        scale *= float(shrink)
    out["winners"] = winners
    return out


def lq_plan(table, x0, Q, R, QN, u_lo, u_hi):
    """The LQ plan (csrc/acmpc_lq.h, round 4): the reference's control QP (control/solvers/control.py:26-79) WITHOUT its box
    rows is a finite-horizon LQ problem with the affine dynamics of linearise() (dynamics.py:65-103); a backward Riccati
    pass gives its optimum as a feedback du_i = K_i x_i + k_i, which is rolled forward from x0 with every control rounded
    to float32 and clipped into the input box as it goes.  Restated line by line in the library's operation order -
    float64, no fused multiply-add: bit-identical to acmpc_lq_plan().  table [7, n] float64, x0 (e_y, e_psi, t) ->
    [n, 2] float32 (v, kappa), or None where the library gives no plan."""
    table = np.asarray(table, dtype=np.float64)
    n = table.shape[1]
    kappa, ds, vel = table[ROW_KAPPA], table[ROW_DS], table[ROW_V]
    Q, R, QN = ([float(x) for x in w] for w in (Q, R, QN))
    gains = np.zeros((n, 8))
    P00, P01, P02, P11, P12, P22 = QN[0], 0.0, 0.0, QN[1], 0.0, QN[2]
    p0 = p1 = p2 = 0.0
    with np.errstate(all="ignore"):
        for i in range(n - 1, -1, -1):
            d = float(ds[i]); ki = float(kappa[i]); vi = float(vel[i])
            a = -(ki * ki) * d
            g = -ki / (vi * d + EPS)
            b = -1.0 / (vi * vi * d + EPS)
            c = 1.0 / (vi * d + EPS)
            h0, h1, h2 = b * P02, b * P12, b * P22
            m0, m1, m2 = d * P01, d * P11, d * P12
            Quu00 = R[0] + b * h2
            Quu01 = d * h1
            Quu11 = R[1] + d * m1
            S00, S01, S02 = (h0 + a * h1) + g * h2, d * h0 + h1, h2
            S10, S11, S12 = (m0 + a * m1) + g * m2, d * m0 + m1, m2
            w0, w1, w2 = c * P02 + p0, c * P12 + p1, c * P22 + p2
            qu0, qu1 = b * w2, d * w1
            det = Quu00 * Quu11 - Quu01 * Quu01
            if not (det > 0.0) or not math.isfinite(det):
                return None
            inv = 1.0 / det
            K00, K01, K02 = (-inv * (Quu11 * S00 - Quu01 * S10), -inv * (Quu11 * S01 - Quu01 * S11),
                             -inv * (Quu11 * S02 - Quu01 * S12))
            K10, K11, K12 = (-inv * (Quu00 * S10 - Quu01 * S00), -inv * (Quu00 * S11 - Quu01 * S01),
                             -inv * (Quu00 * S12 - Quu01 * S02))
            k0, k1 = -inv * (Quu11 * qu0 - Quu01 * qu1), -inv * (Quu00 * qu1 - Quu01 * qu0)
            gains[i] = (K00, K01, K02, k0, K10, K11, K12, k1)
            t00, t10, t20 = (P00 + a * P01) + g * P02, (P01 + a * P11) + g * P12, (P02 + a * P12) + g * P22
            t01, t11, t21 = d * P00 + P01, d * P01 + P11, d * P02 + P12
            t02, t12, t22 = P02, P12, P22
            N00, N01, N02 = (t00 + a * t10) + g * t20, (t01 + a * t11) + g * t21, (t02 + a * t12) + g * t22
            N11, N12, N22 = d * t01 + t11, d * t02 + t12, t22
            n00 = (Q[0] + N00) + (S00 * K00 + S10 * K10)
            n01 = N01 + (S00 * K01 + S10 * K11)
            n02 = N02 + (S00 * K02 + S10 * K12)
            n11 = (Q[1] + N11) + (S01 * K01 + S11 * K11)
            n12 = N12 + (S01 * K02 + S11 * K12)
            n22 = (Q[2] + N22) + (S02 * K02 + S12 * K12)
            q0 = ((w0 + a * w1) + g * w2) + (S00 * k0 + S10 * k1)
            q1 = (d * w0 + w1) + (S01 * k0 + S11 * k1)
            q2 = w2 + (S02 * k0 + S12 * k1)
            P00, P01, P02, P11, P12, P22 = n00, n01, n02, n11, n12, n22
            p0, p1, p2 = q0, q1, q2
        lo, hi = np.asarray(u_lo, dtype=np.float32), np.asarray(u_hi, dtype=np.float32)
        ey, ep, t = (float(x) for x in x0)
        out = np.zeros((n, 2), dtype=np.float32)
        for i in range(n):
            d = float(ds[i]); ki = float(kappa[i]); vi = float(vel[i])
            a = -(ki * ki) * d
            g = -ki / (vi * d + EPS)
            b = -1.0 / (vi * vi * d + EPS)
            c = 1.0 / (vi * d + EPS)
            G = [float(x) for x in gains[i]]
            dv = ((G[0] * ey + G[1] * ep) + G[2] * t) + G[3]
            dk = ((G[4] * ey + G[5] * ep) + G[6] * t) + G[7]
            v = np.fmin(np.fmax(np.float32(vi + dv), lo[0]), hi[0])
            k = np.fmin(np.fmax(np.float32(ki + dk), lo[1]), hi[1])
            out[i] = (v, k)
            cv, ck = float(v) - vi, float(k) - ki
            ey, ep, t = ey + d * ep, (ep + a * ey) + d * ck, ((t + g * ey) + b * cv) + c
    return out if np.isfinite(out).all() else None


# --- the box-constrained LQ plan (csrc/acmpc_lq_box.h, round 5) ---------------------------------------------------------
LQBOX_ALPHA = 1.6
LQBOX_RHO_EY, LQBOX_RHO_T = 3.0e-3, 3.0e-2
LQBOX_WARM_ITERATIONS = 12
LQBOX_PER_TOL = (1.0e4, 1.0e5, 1.0e3, 1.0e6)   # 1 / tolerance of e_y [m], t [s], v [m/s], kappa [1/m]


def _clip64(x, lo, hi):
    # std::fmin(std::fmax(x, lo), hi) on doubles (NaN operands dropped, as IEEE maxNum / minNum)
    return float(np.fmin(np.fmax(x, lo), hi))


def lq_box_rollout_cost(table, x0, Q, R, QN, u_lo, u_hi, margin, plan):
    """csrc/acmpc_lq_box.h rollout_cost(): what the kernels charge a plan, in float64 - tracking cost J, summed squared
    excess V over the state box rows (control.py:57-60,134), the largest |entry| of the decision vector and whether a
    control sits on the input box.  plan [n, 2] float32."""
    table = np.asarray(table, dtype=np.float64)
    n = table.shape[1]
    kappa, ds, width, vel = table[ROW_KAPPA], table[ROW_DS], table[ROW_WIDTH], table[ROW_V]
    Q, R, QN = ([float(x) for x in w] for w in (Q, R, QN))
    lo, hi = np.asarray(u_lo, dtype=np.float32), np.asarray(u_hi, dtype=np.float32)
    plan = np.asarray(plan, dtype=np.float32)
    ey, ep, t = (float(x) for x in x0)
    J = V = 0.0
    biggest = max(abs(ey), abs(ep), abs(t))
    saturated = False
    margin = float(margin)
    with np.errstate(all="ignore"):
        for i in range(n):
            d = float(ds[i]); ki = float(kappa[i]); vi = float(vel[i])
            a = -(ki * ki) * d
            g = -ki / (vi * d + EPS)
            b = -1.0 / (vi * vi * d + EPS)
            c = 1.0 / (vi * d + EPS)
            v, k = plan[i, 0], plan[i, 1]
            saturated = saturated or bool(v <= lo[0] or v >= hi[0] or k <= lo[1] or k >= hi[1])
            dv, dk = float(v) - vi, float(k) - ki
            J += 0.5 * ((((Q[0] * ey) * ey + (Q[1] * ep) * ep) + (Q[2] * t) * t) + ((R[0] * dv) * dv + (R[1] * dk) * dk))
            ey, ep, t = ey + d * ep, (ep + a * ey) + d * dk, ((t + g * ey) + b * dv) + c
            half = float(width[i]) / 2.0 - margin
            over = float(np.fmax(np.fmax(-half - ey, ey - half), 0.0))
            early = float(np.fmax(T_MIN - t, 0.0))
            V += over * over + early * early
            biggest = float(np.fmax(biggest, np.fmax(np.fmax(abs(ey), abs(ep)), np.fmax(abs(t), np.fmax(abs(float(v)), abs(float(k)))))))
        J += 0.5 * (((QN[0] * ey) * ey + (QN[1] * ep) * ep) + (QN[2] * t) * t)
    return dict(J=J, V=V, biggest=biggest, saturated=saturated)


def lq_box_plan(table, x0, Q, R, QN, u_lo, u_hi, margin, w_bound, iterations, state=None):
    """The box-constrained LQ plan (csrc/acmpc_lq_box.h, round 5): the reference's control QP WITH its box rows
    (control/solvers/control.py:26-79,130-144) by the operator splitting of O'Donoghue, Stathopoulos & Boyd (2013) - the
    linearised model (dynamics.py:65-103) stays a hard constraint of the z-update, which is one Riccati factorisation per
    path and a backward vector pass + forward rollout per iteration; the w-update clips into the box rows.  Restated line
    by line in the library's operation order (float64, no fused multiply-add): bit-identical to acmpc_lq_box_plan().
    `state`: None (cold) or the dict a previous call returned under "state" (wx, wu, lx, lu [n, 2] each).
    Returns dict(plan [n, 2] float32 or None, iterations, chosen (0 LQ plan, 1 w iterate, 2 clipped z iterate), triggered,
    J, V, state)."""
    table = np.asarray(table, dtype=np.float64)
    n = table.shape[1]
    kappa, ds, width, vel = table[ROW_KAPPA], table[ROW_DS], table[ROW_WIDTH], table[ROW_V]
    lo32, hi32 = np.asarray(u_lo, dtype=np.float32), np.asarray(u_hi, dtype=np.float32)
    plan = lq_plan(table, x0, Q, R, QN, lo32, hi32)
    if plan is None:
        return dict(plan=None, iterations=0, chosen=0, triggered=False, J=float("nan"), V=float("nan"), state=None)
    Qf, Rf, QNf = ([float(x) for x in w] for w in (Q, R, QN))
    margin = float(margin)
    cost = lq_box_rollout_cost(table, x0, Qf, Rf, QNf, lo32, hi32, margin, plan)
    accept = 1.0e-3 + 1.0e-3 * cost["biggest"]
    triggered = bool(cost["saturated"] or cost["V"] > accept * accept or not (cost["V"] == cost["V"]))
    out = dict(plan=plan, iterations=0, chosen=0, triggered=triggered, J=cost["J"], V=cost["V"], state=None)
    if not triggered or iterations < 1:
        return out
    rho = (LQBOX_RHO_EY, LQBOX_RHO_T, Rf[0], Rf[1])
    if not (rho[2] > 0.0) or not (rho[3] > 0.0):
        return out
    rows = np.zeros((n, 5))
    fac = np.zeros((n, 18))
    with np.errstate(all="ignore"):
        # factor()
        R0, R1 = Rf[0] + rho[2], Rf[1] + rho[3]
        P00, P01, P02, P11, P12, P22 = QNf[0] + rho[0], 0.0, 0.0, QNf[1], 0.0, QNf[2] + rho[1]
        for i in range(n - 1, -1, -1):
            d = float(ds[i]); ki = float(kappa[i]); vi = float(vel[i])
            a = -(ki * ki) * d
            g = -ki / (vi * d + EPS)
            b = -1.0 / (vi * vi * d + EPS)
            c = 1.0 / (vi * d + EPS)
            rows[i] = (d, a, g, b, c)
            h0, h1, h2 = b * P02, b * P12, b * P22
            m0, m1, m2 = d * P01, d * P11, d * P12
            Quu00 = R0 + b * h2
            Quu01 = d * h1
            Quu11 = R1 + d * m1
            S00, S01, S02 = (h0 + a * h1) + g * h2, d * h0 + h1, h2
            S10, S11, S12 = (m0 + a * m1) + g * m2, d * m0 + m1, m2
            det = Quu00 * Quu11 - Quu01 * Quu01
            if not (det > 0.0) or not math.isfinite(det):
                return out
            inv = 1.0 / det
            I00, I01, I11 = inv * Quu11, -(inv * Quu01), inv * Quu00
            K00, K01, K02 = -(I00 * S00 + I01 * S10), -(I00 * S01 + I01 * S11), -(I00 * S02 + I01 * S12)
            K10, K11, K12 = -(I01 * S00 + I11 * S10), -(I01 * S01 + I11 * S11), -(I01 * S02 + I11 * S12)
            fac[i] = (K00, K01, K02, K10, K11, K12, I00, I01, I11, S00, S01, S02, S10, S11, S12, P02, P12, P22)
            t00, t10, t20 = (P00 + a * P01) + g * P02, (P01 + a * P11) + g * P12, (P02 + a * P12) + g * P22
            t01, t11, t21 = d * P00 + P01, d * P01 + P11, d * P02 + P12
            t02, t12, t22 = P02, P12, P22
            N00, N01, N02 = (t00 + a * t10) + g * t20, (t01 + a * t11) + g * t21, (t02 + a * t12) + g * t22
            N11, N12, N22 = d * t01 + t11, d * t02 + t12, t22
            q0 = Qf[0] + rho[0] if i >= 1 else Qf[0]
            q2 = Qf[2] + rho[1] if i >= 1 else Qf[2]
            n00 = (q0 + N00) + (S00 * K00 + S10 * K10)
            n01 = N01 + (S00 * K01 + S10 * K11)
            n02 = N02 + (S00 * K02 + S10 * K12)
            n11 = (Qf[1] + N11) + (S01 * K01 + S11 * K11)
            n12 = N12 + (S01 * K02 + S11 * K12)
            n22 = (q2 + N22) + (S02 * K02 + S12 * K12)
            P00, P01, P02, P11, P12, P22 = n00, n01, n02, n11, n12, n22
        if not (math.isfinite(P00) and math.isfinite(P11) and math.isfinite(P22)):
            return out
        # iterate()
        lo_v, lo_k, hi_v, hi_k = float(lo32[0]), float(lo32[1]), float(hi32[0]), float(hi32[1])
        if state is not None and state["wx"].shape == (n, 2):
            wx, wu, lx, lu = (np.array(state[key], dtype=np.float64) for key in ("wx", "wu", "lx", "lu"))
            iterations = min(iterations, LQBOX_WARM_ITERATIONS)   # a kept iterate continues for at most this many per call
        else:
            wx, wu, lx, lu = (np.zeros((n, 2)) for _ in range(4))
            for i in range(n):
                half = float(width[i]) / 2.0 - margin
                wx[i, 0] = _clip64(0.0, -half, half)
                wx[i, 1] = float(np.fmax(0.0, T_MIN))
                wu[i, 0] = _clip64(0.0, lo_v - float(vel[i]), hi_v - float(vel[i]))
                wu[i, 1] = _clip64(0.0, lo_k - float(kappa[i]), hi_k - float(kappa[i]))
        ks = np.zeros((n, 2))
        zu = np.zeros((n, 2))
        zx = np.zeros((n, 2))
        x0f = [float(x) for x in x0]
        it = 0
        failed = False
        while it < iterations:
            it += 1
            p0 = -(rho[0] * (wx[n - 1, 0] - lx[n - 1, 0]))
            p1 = 0.0
            p2 = -(rho[1] * (wx[n - 1, 1] - lx[n - 1, 1]))
            for i in range(n - 1, -1, -1):
                d, a, g, b, c = (float(x) for x in rows[i])
                F = [float(x) for x in fac[i]]
                w0, w1, w2 = c * F[15] + p0, c * F[16] + p1, c * F[17] + p2
                qu0 = -(rho[2] * (wu[i, 0] - lu[i, 0])) + b * w2
                qu1 = -(rho[3] * (wu[i, 1] - lu[i, 1])) + d * w1
                k0, k1 = -(F[6] * qu0 + F[7] * qu1), -(F[7] * qu0 + F[8] * qu1)
                ks[i] = (k0, k1)
                q0 = q2 = 0.0
                if i >= 1:
                    q0 = -(rho[0] * (wx[i - 1, 0] - lx[i - 1, 0]))
                    q2 = -(rho[1] * (wx[i - 1, 1] - lx[i - 1, 1]))
                n0 = (q0 + ((w0 + a * w1) + g * w2)) + (F[9] * k0 + F[12] * k1)
                n1 = (d * w0 + w1) + (F[10] * k0 + F[13] * k1)
                n2 = (q2 + w2) + (F[11] * k0 + F[14] * k1)
                p0, p1, p2 = float(n0), float(n1), float(n2)
            ey, ep, t = x0f
            gap = move = drift = 0.0
            finite = True
            for i in range(n):
                d, a, g, b, c = (float(x) for x in rows[i])
                F = [float(x) for x in fac[i]]
                dv = ((F[0] * ey + F[1] * ep) + F[2] * t) + float(ks[i, 0])
                dk = ((F[3] * ey + F[4] * ep) + F[5] * t) + float(ks[i, 1])
                ey, ep, t = ey + d * ep, (ep + a * ey) + d * dk, ((t + g * ey) + b * dv) + c
                drift = float(np.fmax(drift, np.fmax(np.fmax(abs(ey - float(zx[i, 0])) * LQBOX_PER_TOL[0], abs(t - float(zx[i, 1])) * LQBOX_PER_TOL[1]),
                                                     np.fmax(abs(dv - float(zu[i, 0])) * LQBOX_PER_TOL[2], abs(dk - float(zu[i, 1])) * LQBOX_PER_TOL[3]))))
                zu[i] = (dv, dk)
                zx[i] = (ey, t)
                half = float(width[i]) / 2.0 - margin
                z = (ey, t, dv, dk)
                blo = (-half, T_MIN, lo_v - float(vel[i]), lo_k - float(kappa[i]))
                bhi = (half, math.inf, hi_v - float(vel[i]), hi_k - float(kappa[i]))
                for q, (w_arr, l_arr, col) in enumerate(((wx, lx, 0), (wx, lx, 1), (wu, lu, 0), (wu, lu, 1))):
                    w_old, l_old = float(w_arr[i, col]), float(l_arr[i, col])
                    relaxed = LQBOX_ALPHA * z[q] + (1.0 - LQBOX_ALPHA) * w_old
                    nxt = _clip64(relaxed + l_old, blo[q], bhi[q])
                    l_new = (l_old + relaxed) - nxt
                    l_arr[i, col] = l_new
                    finite = finite and math.isfinite(relaxed) and math.isfinite(l_new)
                    gap = float(np.fmax(gap, abs(z[q] - nxt) * LQBOX_PER_TOL[q]))
                    move = float(np.fmax(move, abs(nxt - w_old) * LQBOX_PER_TOL[q]))
                    w_arr[i, col] = nxt
            if not finite:
                failed = True
                break
            if move <= 1.0 and (gap <= 1.0 or (it >= 2 and drift <= 1.0)):
                break
        if failed:
            out["iterations"] = -it
            return out
        out["iterations"] = it
        out["state"] = dict(wx=wx, wu=wu, lx=lx, lu=lu)
        best = cost["J"] + float(w_bound) * cost["V"]
        for which, du in ((1, wu), (2, zu)):
            trial = np.zeros((n, 2), dtype=np.float32)
            for i in range(n):
                trial[i, 0] = np.fmin(np.fmax(np.float32(float(vel[i]) + float(du[i, 0])), lo32[0]), hi32[0])
                trial[i, 1] = np.fmin(np.fmax(np.float32(float(kappa[i]) + float(du[i, 1])), lo32[1]), hi32[1])
            if not np.isfinite(trial).all():
                continue
            tc = lq_box_rollout_cost(table, x0, Qf, Rf, QNf, lo32, hi32, margin, trial)
            total = tc["J"] + float(w_bound) * tc["V"]
            if total < best:
                best = total
                out.update(plan=trial, chosen=which, J=tc["J"], V=tc["V"])
    return out


def frenet_start(table, pose):
    """Frenet start state of a pose w.r.t. the path's first waypoint (t2s, dynamics.py:23-40) as csrc/acmpc_lq.h computes it
    for mode T handles: libm cos / sin / fmod on float64 scalars."""
    xr, yr, psir = float(table[ROW_X, 0]), float(table[ROW_Y, 0]), float(table[ROW_PSI, 0])
    x, y, psi = (float(v) for v in pose)
    e_y = math.cos(psir) * (y - yr) - math.sin(psir) * (x - xr)
    wrapped = math.fmod(psi - psir + math.pi, 2.0 * math.pi)
    if wrapped < 0.0:
        wrapped += 2.0 * math.pi
    return np.array([e_y, wrapped - math.pi, 0.0])


# ---------------------------------------------------------------------------
# Particle-filter scoring of the localiser - restates localisation/localiser.py
# ---------------------------------------------------------------------------
def pf_downsample_observation(observation: np.ndarray, average_map_spacing: float) -> np.ndarray:
    """Thin an observed track limit to the map's point spacing (localiser.py:246-253)."""
    spacing = np.mean(np.linalg.norm(observation[1:] - observation[:-1], axis=1))
    n_points = len(observation) * (spacing / average_map_spacing)
    keep = np.zeros(len(observation), dtype=np.bool_)
    keep[np.linspace(0, len(observation) - 1, int(n_points), dtype=np.uint16)] = True
    return observation[keep]


def pf_score_scale(mean: float, sigma: float) -> float:
    """Normaliser of the score: max of the pdf over linspace(-10, 10, 100) (localiser.py:655-661)."""
    x = (np.linspace(-10, 10, 100) - mean) / sigma
    return float(np.max(np.exp(-x**2 / 2.0) / np.sqrt(2.0 * np.pi) / sigma))


def pf_score_particles(states: np.ndarray, centre: np.ndarray, left: np.ndarray, right: np.ndarray,
                       obs_left: np.ndarray, obs_right: np.ndarray, mean: float, sigma: float,
                       thresholds: Dict) -> Dict:
    """_update_particles + _get_valid_particle_mask (localiser.py:255-410,453-462) for P particles.

    states [P,3] float32 (x, y, yaw); centre/left/right map polylines [M,2] float64; obs_* [K,2] float32 track-limit
    points in the vehicle frame (x right, y forward), already downsampled.  Returns the reference's particle dict.
    """
    # 3 nearest-neighbour queries (:282-289)
    offsets, i_centre = nearest_waypoint(states[:, :2].astype(np.float64), centre)
    _, i_left = nearest_waypoint(states[:, :2].astype(np.float64), left)
    _, i_right = nearest_waypoint(states[:, :2].astype(np.float64), right)
    track_indices = np.stack([i_centre, i_left, i_right], axis=1)
    heading = heading_offset(centre, i_centre, states[:, 2])            # (:291-318)
    # observation into every particle's frame (:330-353): only points nearer than 50 m ahead count
    obs_left = obs_left[obs_left[:, 1] < 50]
    obs_right = obs_right[obs_right[:, 1] < 50]
    obs = np.concatenate([obs_left, obs_right])                          # [K,2]
    angle = -states[:, 2] + np.pi / 2
    cos_a, sin_a = np.cos(angle), np.sin(angle)
    # rotation applied is the TRANSPOSE of [[cos, -sin], [sin, cos]] (:355-364)
    x = cos_a[:, None] * obs[None, :, 0] + sin_a[:, None] * obs[None, :, 1]
    y = -sin_a[:, None] * obs[None, :, 0] + cos_a[:, None] * obs[None, :, 1]
    placed = np.stack([x, y], axis=2) + states[:, None, :2]              # [P,K,2]

    def limits(closest, count, track):                                   # (:391-400)
        idx = np.linspace(closest, closest + count, count, dtype=np.uint16)
        return track[np.mod(idx, len(track)).T]

    expected = np.concatenate([limits(i_left, len(obs_left), left), limits(i_right, len(obs_right), right)], axis=1)
    error = np.mean(np.linalg.norm(placed - expected, axis=2), axis=1)   # (:402-410)
    z = (error - mean) / sigma
    score = np.exp(-z**2 / 2.0) / np.sqrt(2.0 * np.pi) / sigma / pf_score_scale(mean, sigma)
    valid = ((heading < thresholds["rotation"]) & (offsets < thresholds["offset"]) & (error < thresholds["track_limit"]))
    return dict(track_indices=track_indices, centreline_idx=i_centre, minimum_offset=offsets, heading_offset=heading,
                observation_error=error, score=score, valid=valid)


def pf_convergence(scores: np.ndarray, states: np.ndarray, max_distance: float, max_angle: float):
    """Weighted-mean estimate and the convergence flag (localiser.py:561-579)."""
    est = estimate_location(scores, states)
    dist = np.linalg.norm(states[:, :2] - est[:2], axis=1)
    ang = np.abs(states[:, 2] - est[2])
    return est, bool(np.max(dist) < max_distance and np.max(ang) < max_angle)


def pf_reset(centre: np.ndarray, n_particles: int):
    """_reset_filter (localiser.py:468-485): particles spread evenly along the centre line, heading along it,
    uniform scores 1/n (float32, as stored in the shared arrays)."""
    idx = np.linspace(0, len(centre) - 3, n_particles).astype(np.int32)
    yaw = np.arctan2(centre[idx + 1, 1] - centre[idx, 1], centre[idx + 1, 0] - centre[idx, 0])
    states = np.vstack((centre[idx, 0], centre[idx, 1], yaw)).T.astype(np.float32)
    scores = np.ones(n_particles, dtype=np.float32)
    scores /= np.sum(scores)
    return states, scores


def pf_resample(states: np.ndarray, scores: np.ndarray, score: np.ndarray, valid: np.ndarray, n_desired: int,
                minimum_particles: int, noise_sigma, rng=np.random):
    """_resample_particles (localiser.py:420-545) on the live particles: keep the valid ones in order; with fewer
    than `minimum_particles` left return None (the caller resets); otherwise top up to `n_desired` with copies of
    valid particles drawn in proportion to `score` (np.random.choice: inverse CDF on uniform draws) plus Gaussian
    noise.  Draw order as in the reference: x noise, y noise, yaw noise, then the indices.
    `states`/`scores` are the float32 published arrays, `score` the float64 scores of this update."""
    kept_states, kept_scores, kept_score = states[valid], scores[valid], score[valid]
    n_valid = kept_states.shape[0]
    if n_valid < minimum_particles:
        return None
    n_new = max(0, n_desired - n_valid)
    noise = np.array([rng.normal(0, noise_sigma[0], n_new), rng.normal(0, noise_sigma[1], n_new),
                      rng.normal(0, noise_sigma[2], n_new)]).T
    with np.errstate(all="ignore"):
        weights = kept_score / np.sum(kept_score)
    if any(np.isnan(weights)):
        weights = np.ones(kept_score.shape) / kept_score.shape[0]
    idx = rng.choice(n_valid, size=n_new, p=weights)
    new_states = np.concatenate((kept_states, kept_states[idx] + noise), axis=0).astype(np.float32)
    new_scores = np.concatenate((kept_scores, kept_scores[idx]), axis=0).astype(np.float32)
    return new_states, new_scores



PF_TAG_RESAMPLE = 0x52534D50  # "RSMP"
PF_TAG_CONTROL = 0x4354524C   # "CTRL"


def pf_weights_fixed_point(score: np.ndarray) -> np.ndarray:
    """floor(score * 2^40) as Python integers for finite positive scores, 0 otherwise (csrc weight_of)."""
    out = []
    for v in np.asarray(score, dtype=np.float64):
        out.append(int(np.floor(v * 2.0**40)) if v > 0.0 else 0)
    return out


def pf_resample_counter_based(states, scores, score, valid, n_desired, minimum_particles, noise_sigma, seed, counter):
    """Restates csrc/acmpc_pf.hip pf_resample_kernel (the device-resident filter's resampling; BUILD-DEFINED draws - the
    reference's own come from NumPy's global stream, see `pf_resample`): keep the valid particles in order; with fewer
    than `minimum_particles` return None (the caller resets); otherwise particle j of the top-up is a copy of kept
    particle  upper_bound(cdf, (r_j * total) >> 64)  with r_j the first 64 bits of Philox4x32-10 at counter
    (j, counter, "RSMP", 0), integer weights floor(score 2^40), plus sigma * Box-Muller normals from the draw at
    (j, counter, "RSMP", 1).  Returns (new_states float32, new_scores float32, picked indices into the kept set) -
    the indices are exact, the noise is float64 here and fast float32 transcendental on the device (~1e-6)."""
    states = np.asarray(states, dtype=np.float32)
    keep = np.flatnonzero(np.asarray(valid, dtype=bool))
    n_valid = keep.shape[0]
    if n_valid < minimum_particles:
        return None
    weights = pf_weights_fixed_point(np.asarray(score, dtype=np.float64)[keep])
    if sum(weights) == 0:
        weights = [1] * n_valid
    cdf = np.cumsum(np.array(weights, dtype=object))
    total = int(cdf[-1])
    n_new = max(0, n_desired - n_valid)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    j = np.arange(n_new, dtype=np.uint32)
    ctr = lambda draw: np.stack([j, np.full_like(j, counter), np.full_like(j, PF_TAG_RESAMPLE), np.full_like(j, draw)], 1)
    r = philox4x32_10(ctr(0), np.broadcast_to(key, (n_new, 2)))
    q = philox4x32_10(ctr(1), np.broadcast_to(key, (n_new, 2)))
    picked = np.empty(n_new, dtype=np.int64)
    cdf_list = [int(c) for c in cdf]
    import bisect
    for i in range(n_new):
        word = (int(r[i, 0]) << 32) | int(r[i, 1])
        picked[i] = bisect.bisect_right(cdf_list, (word * total) >> 64)
    u = uniform_open(q)
    z = np.empty((n_new, 3))
    z0, z1 = box_muller_spec(u[:, 0], u[:, 1])
    z2, _ = box_muller_spec(u[:, 2], u[:, 3])
    z[:, 0], z[:, 1], z[:, 2] = z0, z1, z2
    kept_states = states[keep]
    fresh = (kept_states[picked].astype(np.float64) + z * np.asarray(noise_sigma, dtype=np.float64)).astype(np.float32)
    kept_scores = np.asarray(scores, dtype=np.float32)[keep]
    return (np.concatenate([kept_states, fresh]), np.concatenate([kept_scores, kept_scores[picked]]), picked)
