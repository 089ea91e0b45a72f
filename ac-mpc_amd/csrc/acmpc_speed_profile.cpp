// Host-side solver of the reference's speed-profile QP (src/acmpc/control/solvers/speed_profile.py:26-59)
//
//     minimise 1/2 |v|^2 - v_hi' v     s.t.  a_min <= (v[i+1] - v[i]) / (2 ds[i]) <= a_max,   v_min <= v <= v_hi
//
// which the reference hands to the third-party OSQP package.  Same operator splitting (Stellato et al., "OSQP: an
// operator splitting solver for quadratic programs", 2020: over-relaxed ADMM, per-row step sizes, heavier weight on
// equality rows, residual-balancing step-size updates, the package's default 1e-3 absolute/relative stopping test),
// specialised to this problem's structure: the constraint matrix is [D1; I] with D1 bidiagonal, so the linear
// system of every iteration is symmetric tridiagonal and is solved in O(n) by an LDL' sweep.  That is what makes the
// whole-lap profile of the race start (n ~ 10^4 waypoints, spatial_mpc.py:60-87) as cheap per iteration as the
// 49-point horizon profile of every tick.  Plain C++: no GPU work.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/acmpc.h"

namespace {

struct Tridiagonal {
  std::vector<double> d, e, l;  // diagonal pivots, sub-diagonal, elimination factors

  // K = diag(base) + D1' diag(ra) D1 with D1 rows (-g_i, +g_i)
  void factor(const std::vector<double>& base, const std::vector<double>& ra, const std::vector<double>& g) {
    const int n = static_cast<int>(base.size());
    d.assign(n, 0.0);
    e.assign(n > 1 ? n - 1 : 0, 0.0);
    l.assign(n > 1 ? n - 1 : 0, 0.0);
    for (int i = 0; i < n; ++i) d[i] = base[i];
    for (int i = 0; i + 1 < n; ++i) {
      const double w = ra[i] * g[i] * g[i];
      d[i] += w;
      d[i + 1] += w;
      e[i] = -w;
    }
    for (int i = 0; i + 1 < n; ++i) {
      l[i] = e[i] / d[i];
      d[i + 1] -= l[i] * e[i];
    }
  }

  void solve(std::vector<double>& b) const {
    const int n = static_cast<int>(d.size());
    for (int i = 0; i + 1 < n; ++i) b[i + 1] -= l[i] * b[i];
    for (int i = 0; i < n; ++i) b[i] /= d[i];
    for (int i = n - 2; i >= 0; --i) b[i] -= l[i] * b[i + 1];
  }
};

inline double clamp(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

extern "C" int acmpc_speed_profile_qp(const double* v_hi, const double* ds, int32_t n, double a_min, double a_max,
                                      double v_min, int32_t max_iter, double eps_abs, double eps_rel, double* v,
                                      double* y, int32_t warm_start, int32_t* iterations) {
  if (v_hi == nullptr || ds == nullptr || v == nullptr || y == nullptr || n < 2) return ACMPC_EINVAL;
  const int m = n - 1;  // acceleration rows; then n box rows
  const double sigma = 1e-6, alpha = 1.6;
  double rho = 0.1;
  std::vector<double> g(m), lo_a(m, a_min), hi_a(m, a_max), lo_b(n, v_min), hi_b(v_hi, v_hi + n);
  for (int i = 0; i < m; ++i) g[i] = 1.0 / (2.0 * ds[i]);
  std::vector<double> ra(m), rb(n), base(n);
  Tridiagonal K;
  auto refactor = [&]() {
    for (int i = 0; i < m; ++i) ra[i] = (lo_a[i] == hi_a[i]) ? 1e3 * rho : rho;
    for (int i = 0; i < n; ++i) {
      rb[i] = (lo_b[i] == hi_b[i]) ? 1e3 * rho : rho;
      base[i] = 1.0 + sigma + rb[i];
    }
    K.factor(base, ra, g);
  };
  refactor();

  std::vector<double> x(n, 0.0), ya(m, 0.0), yb(n, 0.0), za(m), zb(n), xt(n), zta(m), ztb(n);
  if (warm_start != 0) {
    for (int i = 0; i < n; ++i) x[i] = v[i];
    for (int i = 0; i < m; ++i) ya[i] = y[i];
    for (int i = 0; i < n; ++i) yb[i] = y[m + i];
  }
  for (int i = 0; i < m; ++i) za[i] = clamp(g[i] * (x[i + 1] - x[i]), lo_a[i], hi_a[i]);
  for (int i = 0; i < n; ++i) zb[i] = clamp(x[i], lo_b[i], hi_b[i]);

  int status = 1;  // 1 = maximum iterations reached, 0 = solved
  int it = 0;
  for (it = 1; it <= max_iter; ++it) {
    // rhs = sigma x - q + A'(rho z - y),  q = -v_hi,  A' w = D1' w_a + w_b
    for (int i = 0; i < n; ++i) xt[i] = sigma * x[i] + v_hi[i] + (rb[i] * zb[i] - yb[i]);
    for (int i = 0; i < m; ++i) {
      const double w = ra[i] * za[i] - ya[i];
      xt[i] -= g[i] * w;
      xt[i + 1] += g[i] * w;
    }
    K.solve(xt);
    for (int i = 0; i < m; ++i) zta[i] = g[i] * (xt[i + 1] - xt[i]);
    for (int i = 0; i < n; ++i) ztb[i] = xt[i];
    for (int i = 0; i < n; ++i) x[i] = alpha * xt[i] + (1.0 - alpha) * x[i];
    for (int i = 0; i < m; ++i) {
      const double mix = alpha * zta[i] + (1.0 - alpha) * za[i];
      const double zn = clamp(mix + ya[i] / ra[i], lo_a[i], hi_a[i]);
      ya[i] += ra[i] * (mix - zn);
      za[i] = zn;
    }
    for (int i = 0; i < n; ++i) {
      const double mix = alpha * ztb[i] + (1.0 - alpha) * zb[i];
      const double zn = clamp(mix + yb[i] / rb[i], lo_b[i], hi_b[i]);
      yb[i] += rb[i] * (mix - zn);
      zb[i] = zn;
    }
    if (it % 10 != 0) continue;
    // residuals (infinity norms) and OSQP's stopping test
    double r_prim = 0, r_dual = 0, s_ax = 0, s_z = 0, s_px = 0, s_aty = 0, s_q = 0;
    for (int i = 0; i < m; ++i) {
      const double ax = g[i] * (x[i + 1] - x[i]);
      r_prim = std::max(r_prim, std::fabs(ax - za[i]));
      s_ax = std::max(s_ax, std::fabs(ax));
      s_z = std::max(s_z, std::fabs(za[i]));
    }
    for (int i = 0; i < n; ++i) {
      r_prim = std::max(r_prim, std::fabs(x[i] - zb[i]));
      s_ax = std::max(s_ax, std::fabs(x[i]));
      s_z = std::max(s_z, std::fabs(zb[i]));
      double aty = yb[i];
      if (i < m) aty -= g[i] * ya[i];
      if (i > 0) aty += g[i - 1] * ya[i - 1];
      r_dual = std::max(r_dual, std::fabs(x[i] - v_hi[i] + aty));
      s_px = std::max(s_px, std::fabs(x[i]));
      s_aty = std::max(s_aty, std::fabs(aty));
      s_q = std::max(s_q, std::fabs(v_hi[i]));
    }
    const double s_prim = std::max(s_ax, s_z), s_dual = std::max(std::max(s_px, s_aty), s_q);
    if (r_prim <= eps_abs + eps_rel * s_prim && r_dual <= eps_abs + eps_rel * s_dual) {
      status = 0;
      break;
    }
    if (it % 50 == 0) {
      const double ratio = std::sqrt((r_prim / std::max(s_prim, 1e-12)) / std::max(r_dual / std::max(s_dual, 1e-12), 1e-12));
      if (ratio > 5.0 || ratio < 0.2) {
        rho = clamp(rho * ratio, 1e-6, 1e6);
        refactor();
      }
    }
  }
  for (int i = 0; i < n; ++i) v[i] = x[i];
  for (int i = 0; i < m; ++i) y[i] = ya[i];
  for (int i = 0; i < n; ++i) y[m + i] = yb[i];
  if (iterations != nullptr) *iterations = std::min(it, static_cast<int>(max_iter));
  return status;
}
