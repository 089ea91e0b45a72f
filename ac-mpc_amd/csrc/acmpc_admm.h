// The speed-profile QP of the reference (src/acmpc/control/solvers/speed_profile.py:26-59)
//
//     minimise 1/2 |v|^2 - v_hi' v     s.t.  a_min <= (v[i+1] - v[i]) / (2 ds[i]) <= a_max,   v_min <= v <= v_hi
//
// solved by the operator splitting of OSQP (Stellato et al., "OSQP: an operator splitting solver for quadratic
// programs", Math. Prog. Comp. 2020: over-relaxed ADMM, per-row step sizes, heavier weight on equality rows,
// residual-balancing step-size updates, the package's 1e-3 absolute / relative stopping test) specialised to this
// problem's structure: the constraint matrix is [D1; I] with D1 bidiagonal, so the linear system of every iteration
// is symmetric tridiagonal.
//
// ONE statement of the algorithm for the host (acmpc_speed_profile_qp: the whole-lap profile of the race start and
// the CPU tests) and for the device (prologue_kernel: the horizon profile of every control tick).  A "team" of T
// workers runs it: on the host T = 1; on the device T = 64, the lanes of one wavefront.  Worker r owns the elements
// r, r + T, r + 2T, ...; element-wise statements touch only the worker's own elements, values of neighbouring
// elements travel through an exchange buffer between two team barriers, norms are max-reductions (exact in any
// order).  Every element therefore sees the same float64 operations in the same order on both sides, which is what
// makes the device iterate bit-identical to the host's.  Both translation units are built with -ffp-contract=off (no
// implicit fused multiply-add); where an update is spelt fma_() it is ONE IEEE fused multiply-add on both sides
// (v_fma_f64 / vfmadd).
//
// The tridiagonal system K x = b is solved in one of two ways, chosen by n alone (so host and device always agree):
//   n >  kPcrMaxN  LDL' sweeps (Thomas): O(n) work, two sequential chains of n steps - the whole-lap profile;
//   n <= kPcrMaxN  parallel cyclic reduction: ceil(log2 n) levels, every element updated independently at each
//                  level from its neighbours i - s and i + s (s = 1, 2, 4, ...) with multipliers that depend only
//                  on K and are computed once per factorisation; an iteration then costs 2 FMAs per element and level
//                  plus one multiply.  In that form a worker keeps everything it owns in local variables (registers
//                  on the device) for the whole solve: an iteration on a wavefront is eight barrier-separated
//                  exchanges of one double per lane (measured: 10 warm iterations of the 49-point horizon took 35 us
//                  as sweeps through LDS arrays, 21 us as cyclic reduction through LDS arrays).
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

// worker-local loops run over K slots: 2 on the device (unrolled, the slots are registers), 128 on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define ACMPC_SLOTS _Pragma("unroll")
#else
#define ACMPC_SLOTS
#endif

namespace acmpc {
namespace admm {

constexpr int kPcrMaxN = 128;     // systems up to this size use parallel cyclic reduction
constexpr int kPcrMaxLevels = 7;  // ceil(log2(kPcrMaxN))

// Doubles of workspace the solver needs for a problem of n points (see Workspace::bind).
ACMPC_HD constexpr int workspace_doubles(int n) { return (n <= kPcrMaxN ? 8 : 16) * n + 8; }

struct Settings {
  double a_min, a_max, v_min;
  int max_iter;
  int check_every;  // the stopping test runs every `check_every` iterations (OSQP's check_termination)
  double eps_abs, eps_rel;
};

struct Workspace {
  // n > kPcrMaxN: the solver's arrays, n doubles each (the m = n - 1 row arrays too)
  double *g, *ra, *rb, *d, *e, *l, *x, *ya, *yb, *za, *zb, *xt;
  // n <= kPcrMaxN: exchange buffers only (values of neighbouring elements between two barriers), n doubles each
  double* ex[8];
  double* red;  // 8 doubles: reductions / broadcast scalars
  ACMPC_HD void bind(double* base, int n) {
    if (n <= kPcrMaxN) {
      for (int q = 0; q < 8; ++q) ex[q] = base + q * n;
      red = base + 8 * n;
      g = ra = rb = d = e = l = x = ya = yb = za = zb = xt = nullptr;
      return;
    }
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
    for (int q = 0; q < 8; ++q) ex[q] = nullptr;
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

// ---------------------------------------------------------------------------------------------------------------------
// n <= kPcrMaxN: cyclic reduction, worker-local state
// ---------------------------------------------------------------------------------------------------------------------
// K = elements per worker: ceil(n / team size) rounded up to what is instantiated (1 or 2 on a wavefront - an
// instruction issued for an empty slot costs the lone wavefront as much as one for a full slot - kPcrMaxN on the host)
template <class Team, int K_>
struct Small {
  static constexpr int K = K_;
  // per owned element (slot k <-> element rank + k * size)
  double g[K], gm[K];            // 1 / (2 ds_i) of row i and of row i - 1
  double ra[K], rb[K], ira[K], irb[K];
  double vh[K];
  double x[K], ya[K], yb[K], za[K], zb[K];
  double pa[kPcrMaxLevels][K], pg[kPcrMaxLevels][K], binv[K];
};

// Everything below is written without data-dependent branches: a worker's slot that holds no element (index >= n)
// computes on clamped indices and simply never stores, and an element without a neighbour at distance s has a zero
// multiplier for it by construction (a_i = 0 for i < s, c_i = 0 for i + s >= n, g = 0 for the row that does not
// exist), so the fused multiply-add with the clamped neighbour adds nothing.  On a lone wavefront every exec-mask
// branch costs as much as the arithmetic it guards (the branching form of these loops was 1 300 instructions per
// iteration, 3 us; this one is a few hundred).
ACMPC_HD int clamp_index(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }

// K = diag(1 + sigma + rb) + D1' diag(ra) D1, D1 rows (-g_i, +g_i), reduced level by level: the equation of element i
//     a_i x_{i-s} + b_i x_i + c_i x_{i+s} = r_i
// sheds its two neighbours by adding alpha_i = -a_i / b_{i-s} times equation i - s and gamma_i = -c_i / b_{i+s} times
// equation i + s; after ceil(log2 n) levels only b_i x_i = r_i is left.  Kept: alpha, gamma of every level, 1 / b_i of
// the last, and the reciprocals of the step sizes (the iteration multiplies by them instead of dividing).
template <class Team, int K>
ACMPC_HD void factor_small(const Team& team, const Workspace& w, Small<Team, K>& st, int n, double rho,
                           const Settings& s) {
  const int m = n - 1;
  double a[K], b[K], c[K];
ACMPC_SLOTS
  for (int k = 0; k < K; ++k) {
    const int i = team.rank() + k * Team::size;
    if (Team::size == 1 && i >= n) break;
    st.ra[k] = (s.a_min == s.a_max) ? 1e3 * rho : rho;
    st.rb[k] = (s.v_min == st.vh[k]) ? 1e3 * rho : rho;
    st.ira[k] = 1.0 / st.ra[k];
    st.irb[k] = 1.0 / st.rb[k];
    // off-diagonal towards i + 1 (row i of D1) and towards i - 1 (row i - 1: the same step size, ra is uniform);
    // g = 0 for the row m that does not exist and gm = 0 for element 0
    const double up = st.ra[k] * st.g[k] * st.g[k];
    const double down = st.ra[k] * st.gm[k] * st.gm[k];
    a[k] = -down;
    b[k] = ((1.0 + kSigma + st.rb[k]) + down) + up;   // the two rank-one terms in the order of the row index
    c[k] = -up;
  }
  (void)m;
ACMPC_SLOTS   // (unrolled on the device: `level` indexes worker-local arrays, which must stay in registers)
  for (int level = 0; level < kPcrMaxLevels; ++level) {
    const int sft = 1 << level;
    if (sft >= n) break;
    double* ea = w.ex[(level & 1) * 3 + 0];
    double* eb = w.ex[(level & 1) * 3 + 1];
    double* ec = w.ex[(level & 1) * 3 + 2];
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      if (i < n) {
        ea[i] = a[k];
        eb[i] = b[k];
        ec[i] = c[k];
      }
    }
    team.sync();
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      if (Team::size == 1 && i >= n) break;
      const int lo = clamp_index(i - sft, n), hi = clamp_index(i + sft, n);
      const double al = -a[k] / eb[lo];   // a_i = 0 when there is no element i - s: alpha = 0
      const double ga = -c[k] / eb[hi];
      st.pa[level][k] = al;
      st.pg[level][k] = ga;
      b[k] = fma_(ga, ea[hi], fma_(al, ec[lo], b[k]));
      a[k] = al * ea[lo];
      c[k] = ga * ec[hi];
    }
    // (the next level writes the other set of buffers; the one after that comes behind the next barrier)
  }
ACMPC_SLOTS
  for (int k = 0; k < K; ++k) st.binv[k] = 1.0 / b[k];
  team.sync();  // the exchange buffers are free again
}

template <class Team, int K>
ACMPC_HD int solve_small(const Team& team, const Workspace& w, const double* v_hi, const double* ds, int n,
                         const Settings& s, double* v, double* y, int warm, int* iterations) {
  const int m = n - 1;  // acceleration rows; then n box rows
  Small<Team, K> st;
  double rho = 0.1;
ACMPC_SLOTS
  for (int k = 0; k < K; ++k) {
    const int i = team.rank() + k * Team::size;
    if (Team::size == 1 && i >= n) break;
    const int ic = clamp_index(i, n);
    // a slot without an element copies element n - 1 and never stores; the last element has no row (g = 0), the first
    // no row behind it (gm = 0)
    st.g[k] = (i < m) ? 1.0 / (2.0 * ds[clamp_index(i, m)]) : 0.0;
    st.gm[k] = (i > 0 && i < n) ? 1.0 / (2.0 * ds[clamp_index(i - 1, m)]) : 0.0;
    st.vh[k] = v_hi[ic];
    st.x[k] = warm != 0 ? v[ic] : 0.0;
    st.yb[k] = warm != 0 ? y[m + ic] : 0.0;
    st.ya[k] = (warm != 0 && i < m) ? y[clamp_index(i, m)] : 0.0;
    st.za[k] = st.zb[k] = 0.0;
  }
  factor_small(team, w, st, n, rho, s);
  // z = clamp(A x): the rows need x_{i+1}
ACMPC_SLOTS
  for (int k = 0; k < K; ++k) {
    const int i = team.rank() + k * Team::size;
    if (i < n) w.ex[6][i] = st.x[k];
  }
  team.sync();
ACMPC_SLOTS
  for (int k = 0; k < K; ++k) {
    const int i = team.rank() + k * Team::size;
    if (Team::size == 1 && i >= n) break;
    const double za = clamp(st.g[k] * (w.ex[6][clamp_index(i + 1, n)] - st.x[k]), s.a_min, s.a_max);
    st.za[k] = (i < m) ? za : 0.0;
    st.zb[k] = clamp(st.x[k], s.v_min, st.vh[k]);
  }

  int status = 1;
  int it = 0;
  for (it = 1; it <= s.max_iter; ++it) {
    // rhs = sigma x - q + A'(rho z - y),  q = -v_hi,  A' u = D1' u_a + u_b: row i's u_a reaches elements i and i + 1
    double ua[K], cur[K];
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      ua[k] = st.ra[k] * st.za[k] - st.ya[k];
      if (i < n) w.ex[7][i] = ua[k];
    }
    team.sync();
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      if (Team::size == 1 && i >= n) break;
      double r = kSigma * st.x[k] + st.vh[k] + (st.rb[k] * st.zb[k] - st.yb[k]);
      r += st.gm[k] * w.ex[7][clamp_index(i - 1, n)];
      r -= st.g[k] * ua[k];
      cur[k] = r;
    }
    // K xt = rhs by cyclic reduction: r_i += alpha_i r_{i-s} + gamma_i r_{i+s} level by level, then xt = r / b
ACMPC_SLOTS
    for (int level = 0; level < kPcrMaxLevels; ++level) {
      const int sft = 1 << level;
      if (sft >= n) break;
      double* buf = w.ex[level & 1];
ACMPC_SLOTS
      for (int k = 0; k < K; ++k) {
        const int i = team.rank() + k * Team::size;
        if (i < n) buf[i] = cur[k];
      }
      team.sync();
ACMPC_SLOTS
      for (int k = 0; k < K; ++k) {
        const int i = team.rank() + k * Team::size;
        if (Team::size == 1 && i >= n) break;
        cur[k] = fma_(st.pg[level][k], buf[clamp_index(i + sft, n)],
                      fma_(st.pa[level][k], buf[clamp_index(i - sft, n)], cur[k]));
      }
    }
    double xt[K];
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      xt[k] = cur[k] * st.binv[k];
      if (i < n) w.ex[2][i] = xt[k];
    }
    team.sync();
    // over-relaxation, projection, dual update
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      if (Team::size == 1 && i >= n) break;
      {
        const double zt = st.g[k] * (w.ex[2][clamp_index(i + 1, n)] - xt[k]);
        const double mix = kAlpha * zt + (1.0 - kAlpha) * st.za[k];
        const double zn = clamp(mix + st.ya[k] * st.ira[k], s.a_min, s.a_max);
        const double ya = st.ya[k] + st.ra[k] * (mix - zn);
        st.ya[k] = (i < m) ? ya : 0.0;   // the row m that does not exist keeps z = y = 0
        st.za[k] = (i < m) ? zn : 0.0;
      }
      const double mix = kAlpha * xt[k] + (1.0 - kAlpha) * st.zb[k];
      const double zn = clamp(mix + st.yb[k] * st.irb[k], s.v_min, st.vh[k]);
      st.yb[k] += st.rb[k] * (mix - zn);
      st.zb[k] = zn;
      st.x[k] = kAlpha * xt[k] + (1.0 - kAlpha) * st.x[k];
    }
    if (it % s.check_every != 0) continue;
    // residuals (infinity norms) and OSQP's stopping test: rows need x_{i+1}, the dual residual ya_{i-1}
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      if (i < n) {
        w.ex[3][i] = st.x[k];
        w.ex[4][i] = st.ya[k];
      }
    }
    team.sync();
    double r_prim = 0, r_dual = 0, s_ax = 0, s_z = 0, s_px = 0, s_aty = 0, s_q = 0;
ACMPC_SLOTS
    for (int k = 0; k < K; ++k) {
      const int i = team.rank() + k * Team::size;
      if (Team::size == 1 && i >= n) break;
      const double live = (i < n) ? 1.0 : 0.0;   // a slot without an element contributes zeros to the norms
      const double ax = st.g[k] * (w.ex[3][clamp_index(i + 1, n)] - st.x[k]);   // 0 for the row that does not exist
      r_prim = dmax(r_prim, live * dabs(ax - st.za[k]));
      s_ax = dmax(s_ax, live * dabs(ax));
      s_z = dmax(s_z, live * dabs(st.za[k]));
      r_prim = dmax(r_prim, live * dabs(st.x[k] - st.zb[k]));
      s_ax = dmax(s_ax, live * dabs(st.x[k]));
      s_z = dmax(s_z, live * dabs(st.zb[k]));
      double aty = st.yb[k];
      aty -= st.g[k] * st.ya[k];
      aty += st.gm[k] * w.ex[4][clamp_index(i - 1, n)];
      r_dual = dmax(r_dual, live * dabs(st.x[k] - st.vh[k] + aty));
      s_px = dmax(s_px, live * dabs(st.x[k]));
      s_aty = dmax(s_aty, live * dabs(aty));
      s_q = dmax(s_q, live * dabs(st.vh[k]));
    }
    r_prim = team.max(r_prim, w.red);
    r_dual = team.max(r_dual, w.red);
    const double s_prim = team.max(dmax(s_ax, s_z), w.red);
    const double s_dual = team.max(dmax(dmax(s_px, s_aty), s_q), w.red);
    if (r_prim <= s.eps_abs + s.eps_rel * s_prim && r_dual <= s.eps_abs + s.eps_rel * s_dual) {
      status = 0;
      break;
    }
    team.sync();  // every worker has read the exchange buffers of the residuals before they are written again
    if (it % 50 == 0) {
      const double ratio =
          __builtin_sqrt((r_prim / dmax(s_prim, 1e-12)) / dmax(r_dual / dmax(s_dual, 1e-12), 1e-12));
      if (ratio > 5.0 || ratio < 0.2) {
        rho = clamp(rho * ratio, 1e-6, 1e6);
        factor_small(team, w, st, n, rho, s);
      }
    }
  }
  team.sync();
ACMPC_SLOTS
  for (int k = 0; k < K; ++k) {
    const int i = team.rank() + k * Team::size;
    if (i < n) {
      v[i] = st.x[k];
      y[m + i] = st.yb[k];
      if (i < m) y[i] = st.ya[k];
    }
  }
  team.sync();
  if (iterations != nullptr) *iterations = it < s.max_iter ? it : s.max_iter;
  return status;
}

// ---------------------------------------------------------------------------------------------------------------------
// n > kPcrMaxN: LDL' sweeps on workspace arrays
// ---------------------------------------------------------------------------------------------------------------------
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

// The QP's EXACT optimum without iterating, when it has the shape every configuration of the reference gives it.
//     min 1/2 |v|^2 - v_hi'v = 1/2 |v - v_hi|^2 + const   s.t.   v_min <= v <= v_hi,   a_min <= (v[i+1] - v[i]) / (2 ds[i]) <= a_max
// The ceiling the objective pulls v towards is the box's own upper bound, so no feasible v exceeds it; and the feasible set
// is closed under the pointwise maximum (every row is a bound, or of the form v[j] <= v[k] + c), so it has a pointwise
// LARGEST element v*, which is then nearer to v_hi than any other feasible v in every coordinate: the optimum.  With
// a_min <= 0 <= a_max, v* is v_hi cut down by a forward pass (v[i+1] <= v[i] + 2 ds[i] a_max: no faster out of a slow
// point than a_max allows) and a backward pass (v[i] <= v[i+1] - 2 ds[i] a_min: no faster into one than braking allows) -
// the tightest bounds a chain of difference rows implies; a path that goes forward and comes back only adds non-negative
// terms.  The QP is feasible exactly when v* >= v_min everywhere.  OSQP - and its restatement below - get to within their
// tolerance of this point in 5 iterations from the previous tick's iterate on a steady path and in 100 - 400 where the
// car approaches a braking zone (the closed loop's p90 tick was 170 us, p99 300 us for it; round 5).
//
// Each pass is v[i] = min over j of (v_hi[j] + the gaps between j and i), a prefix "minimum of sums" - evaluated as a scan
// with doubling distances (Hillis - Steele), which is the SPECIFICATION of the roundings on both sides:
//     forward    v[i] = v_hi[i],  G[i] = u[i - 1] = (2 ds[i - 1]) a_max  (G[0] = 0)
//                for d = 1, 2, 4, ... < n, every i >= d at once:   v[i] = min(v[i], v[i - d] + G[i]),   G[i] = G[i] + G[i - d]
//     backward   the same towards lower indices with b[i] = (-2 ds[i]) a_min, from the forward pass's result
// (G[i] is the sum of the d gaps that end at i; elements without a partner at distance d keep their values.)  On the
// device a wavefront holds an element per lane and a pass is six exchanges (n <= 64; through the workspace for more);
// as two serial sweeps - 2 n dependent steps - the same profile took a lone wavefront 5.3 us at n = 49.
// Returns true with v = the optimum and y = 0 (no dual iterate: nothing iterates), on every worker alike; false - v and y
// untouched or partly written - when the problem is not of this shape (a_min > 0, a_max < 0, a non-finite or non-positive
// spacing, a non-finite ceiling) or infeasible: the caller then runs the splitting, whose status is the reference's own
// for such a problem.  `scratch`: 4 n doubles (not needed by a 64-wide team with n <= 64).  v must not overlap v_hi or ds.
template <class Team>
ACMPC_HD bool exact_profile(const Team& team, const Workspace& w, const double* v_hi, const double* ds, int n,
                            const Settings& s, double* v, double* y, double* scratch) {
  if (!(s.a_max >= 0.0) || !(s.a_min <= 0.0)) return false;   // (the same on every worker)
  const double inf = __builtin_huge_val();
  bool done = false;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (Team::size == 64) {
    if (n <= 64) {   // an element per lane, in registers
      const int i = team.rank();
      const bool mine = i < n;
      double val = mine ? v_hi[i] : inf;
      double gap = (mine && i > 0) ? (2.0 * ds[i - 1]) * s.a_max : 0.0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double other = __shfl_up(val, d, 64), other_gap = __shfl_up(gap, d, 64);
        if (i >= d && d < n) {
          const double cand = other + gap;
          val = cand < val ? cand : val;
          gap = gap + other_gap;
        }
      }
      gap = (i + 1 < n) ? (-2.0 * ds[i]) * s.a_min : 0.0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double other = __shfl_down(val, d, 64), other_gap = __shfl_down(gap, d, 64);
        if (i + d < n) {
          const double cand = other + gap;
          val = cand < val ? cand : val;
          gap = gap + other_gap;
        }
      }
      if (mine) v[i] = val;
      done = true;
    }
  }
#endif
  if (!done) {   // arrays in the workspace, two buffers per pass; every worker takes the elements rank, rank + size, ...
    double* cur_v = scratch;
    double* cur_g = scratch + n;
    double* nxt_v = scratch + 2 * n;
    double* nxt_g = scratch + 3 * n;
    for (int i = team.rank(); i < n; i += Team::size) {
      cur_v[i] = v_hi[i];
      cur_g[i] = i > 0 ? (2.0 * ds[i - 1]) * s.a_max : 0.0;
    }
    team.sync();
    for (int d = 1; d < n; d <<= 1) {
      for (int i = team.rank(); i < n; i += Team::size) {
        double val = cur_v[i], gap = cur_g[i];
        if (i >= d) {
          const double cand = cur_v[i - d] + gap;
          val = cand < val ? cand : val;
          gap = gap + cur_g[i - d];
        }
        nxt_v[i] = val;
        nxt_g[i] = gap;
      }
      team.sync();
      double* t = cur_v; cur_v = nxt_v; nxt_v = t;
      t = cur_g; cur_g = nxt_g; nxt_g = t;
    }
    for (int i = team.rank(); i < n; i += Team::size) cur_g[i] = i + 1 < n ? (-2.0 * ds[i]) * s.a_min : 0.0;
    team.sync();
    for (int d = 1; d < n; d <<= 1) {
      for (int i = team.rank(); i < n; i += Team::size) {
        double val = cur_v[i], gap = cur_g[i];
        if (i + d < n) {
          const double cand = cur_v[i + d] + gap;
          val = cand < val ? cand : val;
          gap = gap + cur_g[i + d];
        }
        nxt_v[i] = val;
        nxt_g[i] = gap;
      }
      team.sync();
      double* t = cur_v; cur_v = nxt_v; nxt_v = t;
      t = cur_g; cur_g = nxt_g; nxt_g = t;
    }
    for (int i = team.rank(); i < n; i += Team::size) v[i] = cur_v[i];
  }
  team.sync();
  double bad = 0.0;
  for (int i = team.rank(); i < n; i += Team::size) {
    if (!(v[i] >= s.v_min) || !(v[i] <= v_hi[i])) bad = 1.0;                       // infeasible, or a non-finite ceiling
    if (i + 1 < n && !(ds[i] > 0.0 && ds[i] < __builtin_huge_val())) bad = 1.0;   // not a path
  }
  bad = team.max(bad, w.red);
  if (bad != 0.0) return false;
  for (int i = team.rank(); i < 2 * n - 1; i += Team::size) y[i] = 0.0;
  team.sync();
  return true;
}

// `v` [n] and `y` [2n - 1] hold the primal / dual iterate: read when warm != 0, always written.  Returns 0 = solved,
// 1 = maximum iterations reached; *iterations = iterations run.
template <class Team>
ACMPC_HD int solve(const Team& team, const Workspace& w, const double* v_hi, const double* ds, int n,
                   const Settings& s, double* v, double* y, int warm, int* iterations) {
  if (n <= kPcrMaxN) {
    constexpr int kSlots = (kPcrMaxN + Team::size - 1) / Team::size;
    if constexpr (kSlots > 1 && Team::size > 1) {
      if (n <= Team::size) return solve_small<Team, 1>(team, w, v_hi, ds, n, s, v, y, warm, iterations);
    }
    return solve_small<Team, kSlots>(team, w, v_hi, ds, n, s, v, y, warm, iterations);
  }
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
