// The speed-profile QP of the reference (src/acmpc/control/solvers/speed_profile.py:26-59)
//
//     minimise 1/2 |v|^2 - v_hi' v     s.t.  a_min <= (v[i+1] - v[i]) / (2 ds[i]) <= a_max,   v_min <= v <= v_hi
//
// solved by the operator splitting of OSQP (Stellato et al., "OSQP: an operator splitting solver for quadratic
// programs", Math. Prog. Comp. 2020: over-relaxed ADMM, per-row step sizes, heavier weight on equality rows,
// residual-balancing step-size updates, the package's 1e-3 absolute / relative stopping test) specialised to this
// problem's structure: the constraint matrix is [D1; I] with D1 bidiagonal, so the linear system of every iteration
// is symmetric tridiagonal and is solved in O(n) by an LDL' sweep.
//
// ONE statement of the algorithm for the host (acmpc_speed_profile_qp: the whole-lap profile of the race start and
// the CPU tests) and for the device (prologue_kernel: the horizon profile of every control tick, inside the solve's
// hipGraph).  A "team" of T workers runs it: on the host T = 1; on the device T = 64, the lanes of one wavefront.
// Element-wise statements are split over the team, the two sequential sweeps of the tridiagonal solve are run by
// worker 0, norms are max-reductions (exact in any order).  Every element therefore sees the same float64 operations
// in the same order on both sides, which is what makes the device iterate bit-identical to the host's.  Both
// translation units are built with -ffp-contract=off (no implicit fused multiply-add); the three sequential sweeps
// (LDL' factorisation, forward and backward substitution) spell their update as ONE fused multiply-add per step
// (fma_() below: IEEE fma, v_fma_f64 on the device, vfmadd on the host).
//
// The tridiagonal system K x = b is solved in one of two ways, chosen by n alone (so host and device always agree):
//   n >  kPcrMaxN  LDL' sweeps (Thomas): O(n) work, two sequential chains of n steps - the whole-lap profile;
//   n <= kPcrMaxN  parallel cyclic reduction: ceil(log2 n) levels, every element updated independently at each
//                  level from its neighbours i - s and i + s (s = 1, 2, 4, ...) with multipliers that depend only
//                  on K and are therefore computed once per factorisation; an iteration then costs 2 FMAs per
//                  element and level plus one multiply.  On a wavefront that is ~7 short parallel phases instead of
//                  two 50-step dependent chains (measured: the horizon's 10 warm iterations took 35 us as sweeps).
// Both are exact solvers of the same SPD, diagonally dominant system; their roundings differ, which is why the
// choice is part of the specification and not a launch decision.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ACMPC_HD __host__ __device__ __forceinline__
#else
#define ACMPC_HD inline
#endif

namespace acmpc {
namespace admm {

constexpr int kPcrMaxN = 128;   // systems up to this size use parallel cyclic reduction
constexpr int kPcrMaxLevels = 7;  // ceil(log2(kPcrMaxN))

// Doubles of workspace the solver needs for a problem of n points (see Workspace::bind).
ACMPC_HD constexpr int workspace_doubles(int n) { return (n <= kPcrMaxN ? 36 : 16) * n; }

struct Settings {
  double a_min, a_max, v_min;
  int max_iter;
  int check_every;  // the stopping test runs every `check_every` iterations (OSQP's check_termination)
  double eps_abs, eps_rel;
};

struct Workspace {
  double *g, *ra, *rb, *d, *e, *l, *x, *ya, *yb, *za, *zb, *xt;  // n doubles each (the m = n - 1 row arrays too)
  double* red;                                                    // >= 8 doubles: reductions / broadcast scalars
  // cyclic reduction (n <= kPcrMaxN only): multipliers of every level, reciprocal of the final diagonal, scratch
  double *pa, *pg, *binv, *tmp, *ca, *cb, *cc, *na, *nb, *nc;
  ACMPC_HD void bind(double* base, int n) {
    g = base;
    ra = base + n;
    rb = base + 2 * n;
    d = base + 3 * n;
    e = base + 4 * n;
    l = base + 5 * n;
    x = base + 6 * n;
    ya = base + 7 * n;
    yb = base + 8 * n;
    za = base + 9 * n;
    zb = base + 10 * n;
    xt = base + 11 * n;
    red = base + 12 * n;
    pa = pg = binv = tmp = ca = cb = cc = na = nb = nc = nullptr;
    if (n <= kPcrMaxN) {
      pa = base + 14 * n;                       // [kPcrMaxLevels][n]
      pg = pa + kPcrMaxLevels * n;              // [kPcrMaxLevels][n]
      binv = pg + kPcrMaxLevels * n;
      tmp = binv + n;
      ca = tmp + n;
      cb = ca + n;
      cc = cb + n;
      na = cc + n;
      nb = na + n;
      nc = nb + n;
    }
  }
};

ACMPC_HD double clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
ACMPC_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
ACMPC_HD double dmax(double a, double b) { return a > b ? a : b; }
ACMPC_HD double dabs(double a) { return a < 0.0 ? -a : a; }

// What a team has to provide: its size, the worker's rank, a barrier that also makes the workspace writes of every
// worker visible to the others, and a max-reduction that returns the team-wide maximum to every worker.
struct HostTeam {
  static constexpr int size = 1;
  ACMPC_HD int rank() const { return 0; }
  ACMPC_HD void sync() const {}
  ACMPC_HD double max(double v, double*) const { return v; }
};

constexpr double kSigma = 1e-6;
constexpr double kAlpha = 1.6;

// K = diag(1 + sigma + rb) + D1' diag(ra) D1, D1 rows (-g_i, +g_i); LDL' factors into d (pivots), l (multipliers)
template <class Team>
ACMPC_HD void refactor(const Team& team, const Workspace& w, int n, double rho, const double* v_hi,
                       const Settings& s) {
  const int m = n - 1;
  for (int i = team.rank(); i < n; i += Team::size) {
    if (i < m) w.ra[i] = (s.a_min == s.a_max) ? 1e3 * rho : rho;
    w.rb[i] = (s.v_min == v_hi[i]) ? 1e3 * rho : rho;
  }
  team.sync();
  for (int i = team.rank(); i < n; i += Team::size) {
    double di = 1.0 + kSigma + w.rb[i];
    // the two rank-one terms arrive in the order of the row index, as a sweep over the rows adds them
    if (i > 0) di += w.ra[i - 1] * w.g[i - 1] * w.g[i - 1];
    if (i < m) {
      const double wi = w.ra[i] * w.g[i] * w.g[i];
      di += wi;
      w.e[i] = -wi;
    }
    w.d[i] = di;
  }
  team.sync();
  if (n <= kPcrMaxN) {
    // cyclic reduction of K (diagonal d, off-diagonals e): level by level the equation of element i
    //     a_i x_{i-s} + b_i x_i + c_i x_{i+s} = r_i
    // sheds its two neighbours by adding alpha_i = -a_i / b_{i-s} times equation i - s and gamma_i = -c_i / b_{i+s}
    // times equation i + s; after ceil(log2 n) levels only b_i x_i = r_i is left.  Kept: alpha, gamma of every level
    // and 1 / b_i of the last.
    for (int i = team.rank(); i < n; i += Team::size) {
      w.ca[i] = (i > 0) ? w.e[i - 1] : 0.0;
      w.cb[i] = w.d[i];
      w.cc[i] = (i < m) ? w.e[i] : 0.0;
    }
    team.sync();
    double *a = w.ca, *b = w.cb, *c = w.cc, *a2 = w.na, *b2 = w.nb, *c2 = w.nc;
    int level = 0;
    for (int s = 1; s < n; s <<= 1, ++level) {
      for (int i = team.rank(); i < n; i += Team::size) {
        const bool lo = i - s >= 0, hi = i + s < n;
        const double al = lo ? -a[i] / b[i - s] : 0.0;
        const double ga = hi ? -c[i] / b[i + s] : 0.0;
        w.pa[level * n + i] = al;
        w.pg[level * n + i] = ga;
        a2[i] = lo ? al * a[i - s] : 0.0;
        c2[i] = hi ? ga * c[i + s] : 0.0;
        double bi = b[i];
        if (lo) bi = fma_(al, c[i - s], bi);
        if (hi) bi = fma_(ga, a[i + s], bi);
        b2[i] = bi;
      }
      team.sync();
      double* t = a; a = a2; a2 = t;
      t = b; b = b2; b2 = t;
      t = c; c = c2; c2 = t;
    }
    for (int i = team.rank(); i < n; i += Team::size) w.binv[i] = 1.0 / b[i];
    team.sync();
    return;
  }
  if (team.rank() == 0) {
    // l[i] = e[i] / d[i]; d[i + 1] = fma(-l[i], e[i], d[i + 1]) - with the running pivot in a register, and the
    // arrays behind restrict-qualified pointers so that their loads do not wait for the stores of earlier steps
    const double* __restrict__ e = w.e;
    double* __restrict__ l = w.l;
    double* __restrict__ d = w.d;
    double pivot = d[0];
#pragma unroll 4
    for (int i = 0; i + 1 < n; ++i) {
      const double li = e[i] / pivot;
      l[i] = li;
      pivot = fma_(-li, e[i], d[i + 1]);
      d[i + 1] = pivot;
    }
  }
  team.sync();
}

// `v` [n] and `y` [2n - 1] hold the primal / dual iterate: read when warm != 0, always written.  Returns 0 = solved,
// 1 = maximum iterations reached; *iterations = iterations run.
template <class Team>
ACMPC_HD int solve(const Team& team, const Workspace& w, const double* v_hi, const double* ds, int n,
                   const Settings& s, double* v, double* y, int warm, int* iterations) {
  const int m = n - 1;  // acceleration rows; then n box rows
  double rho = 0.1;
  for (int i = team.rank(); i < m; i += Team::size) w.g[i] = 1.0 / (2.0 * ds[i]);
  team.sync();
  refactor(team, w, n, rho, v_hi, s);
  for (int i = team.rank(); i < n; i += Team::size) {
    w.x[i] = warm != 0 ? v[i] : 0.0;
    w.yb[i] = warm != 0 ? y[m + i] : 0.0;
    if (i < m) w.ya[i] = warm != 0 ? y[i] : 0.0;
  }
  team.sync();
  for (int i = team.rank(); i < n; i += Team::size) {
    if (i < m) w.za[i] = clamp(w.g[i] * (w.x[i + 1] - w.x[i]), s.a_min, s.a_max);
    w.zb[i] = clamp(w.x[i], s.v_min, v_hi[i]);
  }
  team.sync();

  int status = 1;
  int it = 0;
  for (it = 1; it <= s.max_iter; ++it) {
    // rhs = sigma x - q + A'(rho z - y),  q = -v_hi,  A' u = D1' u_a + u_b
    for (int i = team.rank(); i < n; i += Team::size) {
      double r = kSigma * w.x[i] + v_hi[i] + (w.rb[i] * w.zb[i] - w.yb[i]);
      if (i > 0) r += w.g[i - 1] * (w.ra[i - 1] * w.za[i - 1] - w.ya[i - 1]);
      if (i < m) r -= w.g[i] * (w.ra[i] * w.za[i] - w.ya[i]);
      w.xt[i] = r;
    }
    team.sync();
    if (n <= kPcrMaxN) {
      // K xt = rhs by cyclic reduction: r_i += alpha_i r_{i-s} + gamma_i r_{i+s} level by level, then xt = r / b
      double *cur = w.xt, *nxt = w.tmp;
      int level = 0;
      for (int s2 = 1; s2 < n; s2 <<= 1, ++level) {
        const double* __restrict__ al = w.pa + level * n;
        const double* __restrict__ ga = w.pg + level * n;
        for (int i = team.rank(); i < n; i += Team::size) {
          double r = cur[i];
          if (i - s2 >= 0) r = fma_(al[i], cur[i - s2], r);
          if (i + s2 < n) r = fma_(ga[i], cur[i + s2], r);
          nxt[i] = r;
        }
        team.sync();
        double* t = cur; cur = nxt; nxt = t;
      }
      for (int i = team.rank(); i < n; i += Team::size) {
        const double xi = cur[i] * w.binv[i];
        nxt[i] = xi;          // both buffers end up holding the solution: whichever of them is `xt` is right
        cur[i] = xi;
      }
      team.sync();
    } else {
    // K xt = rhs: forward sweep, pivots, backward sweep
    // (xt[i + 1] = fma(-l[i], xt[i], xt[i + 1]) and xt[i] = fma(-l[i], xt[i + 1], xt[i]), the running value kept in
    // a register: only it is on the dependent chain, the loads of l and xt are not)
    if (team.rank() == 0) {
      const double* __restrict__ l = w.l;
      double* __restrict__ xt = w.xt;
      double run = xt[0];
#pragma unroll 8
      for (int i = 0; i + 1 < n; ++i) {
        run = fma_(-l[i], run, xt[i + 1]);
        xt[i + 1] = run;
      }
    }
    team.sync();
    for (int i = team.rank(); i < n; i += Team::size) w.xt[i] /= w.d[i];
    team.sync();
    if (team.rank() == 0) {
      const double* __restrict__ l = w.l;
      double* __restrict__ xt = w.xt;
      double run = xt[n - 1];
#pragma unroll 8
      for (int i = n - 2; i >= 0; --i) {
        run = fma_(-l[i], run, xt[i]);
        xt[i] = run;
      }
    }
    team.sync();
    }
    // over-relaxation, projection, dual update (rows first read their neighbours' xt, then every x is replaced)
    for (int i = team.rank(); i < m; i += Team::size) {
      const double zt = w.g[i] * (w.xt[i + 1] - w.xt[i]);
      const double mix = kAlpha * zt + (1.0 - kAlpha) * w.za[i];
      const double zn = clamp(mix + w.ya[i] / w.ra[i], s.a_min, s.a_max);
      w.ya[i] += w.ra[i] * (mix - zn);
      w.za[i] = zn;
    }
    for (int i = team.rank(); i < n; i += Team::size) {
      const double mix = kAlpha * w.xt[i] + (1.0 - kAlpha) * w.zb[i];
      const double zn = clamp(mix + w.yb[i] / w.rb[i], s.v_min, v_hi[i]);
      w.yb[i] += w.rb[i] * (mix - zn);
      w.zb[i] = zn;
      w.x[i] = kAlpha * w.xt[i] + (1.0 - kAlpha) * w.x[i];
    }
    team.sync();
    if (it % s.check_every != 0) continue;
    // residuals (infinity norms) and OSQP's stopping test
    double r_prim = 0, r_dual = 0, s_ax = 0, s_z = 0, s_px = 0, s_aty = 0, s_q = 0;
    for (int i = team.rank(); i < n; i += Team::size) {
      if (i < m) {
        const double ax = w.g[i] * (w.x[i + 1] - w.x[i]);
        r_prim = dmax(r_prim, dabs(ax - w.za[i]));
        s_ax = dmax(s_ax, dabs(ax));
        s_z = dmax(s_z, dabs(w.za[i]));
      }
      r_prim = dmax(r_prim, dabs(w.x[i] - w.zb[i]));
      s_ax = dmax(s_ax, dabs(w.x[i]));
      s_z = dmax(s_z, dabs(w.zb[i]));
      double aty = w.yb[i];
      if (i < m) aty -= w.g[i] * w.ya[i];
      if (i > 0) aty += w.g[i - 1] * w.ya[i - 1];
      r_dual = dmax(r_dual, dabs(w.x[i] - v_hi[i] + aty));
      s_px = dmax(s_px, dabs(w.x[i]));
      s_aty = dmax(s_aty, dabs(aty));
      s_q = dmax(s_q, dabs(v_hi[i]));
    }
    r_prim = team.max(r_prim, w.red);
    r_dual = team.max(r_dual, w.red);
    const double s_prim = team.max(dmax(s_ax, s_z), w.red);
    const double s_dual = team.max(dmax(dmax(s_px, s_aty), s_q), w.red);
    if (r_prim <= s.eps_abs + s.eps_rel * s_prim && r_dual <= s.eps_abs + s.eps_rel * s_dual) {
      status = 0;
      break;
    }
    if (it % 50 == 0) {
      const double ratio =
          __builtin_sqrt((r_prim / dmax(s_prim, 1e-12)) / dmax(r_dual / dmax(s_dual, 1e-12), 1e-12));
      if (ratio > 5.0 || ratio < 0.2) {
        rho = clamp(rho * ratio, 1e-6, 1e6);
        refactor(team, w, n, rho, v_hi, s);
      }
    }
  }
  for (int i = team.rank(); i < n; i += Team::size) {
    v[i] = w.x[i];
    y[m + i] = w.yb[i];
    if (i < m) y[i] = w.ya[i];
  }
  team.sync();
  if (iterations != nullptr) *iterations = it < s.max_iter ? it : s.max_iter;
  return status;
}

}  // namespace admm
}  // namespace acmpc
