// The box-constrained LQ plan: the reference's control QP WITH its box rows, on the host (round 5).
//
// acmpc_lq.h gives the optimum of the QP of control/solvers/control.py:26-79 without its box rows and clips the controls
// as it rolls them out.  Where a box row is active far ahead - a corner tighter than the steering box allows, a corridor
// (control.py:57-60 under the racing widths linspace(10, 6, H), controller.py:256-267) the unconstrained line leaves - that
// plan reacts when it gets there; the QP's optimum prepares for it (turns in early, runs wide on entry).  This header
// solves the QP itself: the operator splitting OSQP is built on (Stellato et al. 2020), in the form O'Donoghue,
// Stathopoulos & Boyd give it for optimal control ("A splitting method for optimal control", IEEE TCST 2013) -
//     minimise  1/2 (z - z_ref)' P (z - z_ref) + I_dynamics(z) + I_box(w)    subject to  z = w
// where the z-update keeps the linearised bicycle model (dynamics.py:65-103) as HARD equality rows and is therefore an LQ
// problem with shifted weights and linear terms - one Riccati factorisation per path, then a backward vector pass and a
// forward rollout per iteration, O(n) - and the w-update is a clip into the box rows.  Boxed components, as in
// control.py:47-70,130-144: e_y of x_1 .. x_n (the corridor), t of x_1 .. x_n (>= 0.01), v and kappa of every step.
//     z+ = argmin_dynamics  cost(z) + sum_j rho_j / 2 (z_j - w_j + l_j)^2          (l = the scaled dual)
//     w+ = clip(alpha z+ + (1 - alpha) w + l),   l+ = l + alpha z+ + (1 - alpha) w - w+      (alpha = 1.6)
// rho per component class (e_y, t, v, kappa): the input classes take the cost's own weights r_term, the state classes
// 3e-3 and 3e-2 (swept over 64 constraint-active racing scenarios x 4 tracks' weights: 15-20 iterations to 1e-3 of the
// optimum's tracking cost).  The iterate (w, l) is kept between calls: the next tick's QP is this
// one moved by centimetres, and a converged iterate confirms itself in one or two iterations.
//
// The result is a CANDIDATE, never trusted: the plan handed to the last sampling round is whichever of {LQ plan, the
// box iterate w, the dynamics iterate z clipped} rolls out cheapest under the cost the kernels charge (J + w_bound V),
// and the argmin over the round keeps it only when it wins.
//
// Host only, float64, one fixed operation order, no fused multiply-add: oracle/acmpc_oracle.py lq_box_plan() restates it
// line by line and is bit-identical (tests/test_lq_box.py).
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

#include "acmpc_lq.h"

namespace acmpc {
namespace lqbox {

constexpr double kTMin = 0.01;        // control.py:134
constexpr double kAlpha = 1.6;        // over-relaxation (OSQP's default)
constexpr double kRhoEy = 3.0e-3;     // step sizes of the state classes (the input classes use r_term)
constexpr double kRhoT = 3.0e-2;
#ifndef ACMPC_LQ_BOX_WARM
#define ACMPC_LQ_BOX_WARM 12
#endif
constexpr int kWarmIterations = ACMPC_LQ_BOX_WARM;    // per call from a kept iterate (a cold start takes the caller's cap)
// the iteration stops when every boxed component's z and w agree AND w has stopped moving (or both have stopped moving
// apart: an infeasible row), per class:
// e_y 1e-4 m, t 1e-5 s, v 1e-3 m/s, kappa 1e-6 1/m
constexpr double kPerTolEy = 1.0e4, kPerTolT = 1.0e5, kPerTolV = 1.0e3, kPerTolK = 1.0e6;   // 1 / tolerance

struct State {                 // the iterate kept between calls (one per problem)
  int n = 0;                   // 0: cold
  std::vector<double> wx, wu, lx, lu;   // [n][2] each: (e_y, t) of x_{i+1}; (dv, dkappa) of step i
  void reset() { n = 0; }
};

struct Workspace {             // per-path factorisation + scratch (no allocation in the iteration)
  std::vector<double> rows;    // [n][5]: d, a, g, b, c of linearise()
  std::vector<double> fac;     // [n][18]: K (6), Quu^-1 (3: 00 01 11), S = B'PA (6), third column of P_{i+1} (3)
  std::vector<double> ks;      // [n][2]
  std::vector<double> zx, zu;  // [n][2] each: the dynamics iterate
  std::vector<float> trial;    // [n][2]
};

struct Cost { double J = 0.0, V = 0.0, biggest = 0.0; bool saturated = false; };

// What the kernels charge a plan (csrc/acmpc_device.h step_spatial, oracle rollout_spatial) in float64: tracking cost J,
// summed squared box excess V of the state rows (the plan's controls are inside the input box by construction), the
// largest |entry| of the decision vector, and whether any control sits ON the input box.
inline Cost rollout_cost(const double* table, int n, const double x0[3], const double Q[3], const double R[2],
                         const double QN[3], const float u_lo[2], const float u_hi[2], double margin, const float* plan) {
  const double* kappa = table + 3 * static_cast<size_t>(n);
  const double* ds = table + 4 * static_cast<size_t>(n);
  const double* width = table + 5 * static_cast<size_t>(n);
  const double* vel = table + 6 * static_cast<size_t>(n);
  Cost out;
  double ey = x0[0], ep = x0[1], t = x0[2];
  out.biggest = std::fmax(std::fabs(ey), std::fmax(std::fabs(ep), std::fabs(t)));
  for (int i = 0; i < n; ++i) {
    const double d = ds[i];
    const double a = -(kappa[i] * kappa[i]) * d;
    const double g = -kappa[i] / (vel[i] * d + lq::kEps);
    const double b = -1.0 / (vel[i] * vel[i] * d + lq::kEps);
    const double c = 1.0 / (vel[i] * d + lq::kEps);
    const float v = plan[2 * i], k = plan[2 * i + 1];
    out.saturated = out.saturated || v <= u_lo[0] || v >= u_hi[0] || k <= u_lo[1] || k >= u_hi[1];
    const double dv = static_cast<double>(v) - vel[i], dk = static_cast<double>(k) - kappa[i];
    out.J += 0.5 * ((((Q[0] * ey) * ey + (Q[1] * ep) * ep) + (Q[2] * t) * t) + ((R[0] * dv) * dv + (R[1] * dk) * dk));
    const double ey_n = ey + d * ep;
    const double ep_n = (ep + a * ey) + d * dk;
    const double t_n = ((t + g * ey) + b * dv) + c;
    ey = ey_n, ep = ep_n, t = t_n;
    const double half = width[i] / 2.0 - margin;
    const double over = std::fmax(std::fmax(-half - ey, ey - half), 0.0);
    const double early = std::fmax(kTMin - t, 0.0);
    out.V += over * over + early * early;
    out.biggest = std::fmax(out.biggest, std::fmax(std::fmax(std::fabs(ey), std::fabs(ep)),
                                                   std::fmax(std::fabs(t), std::fmax(std::fabs(static_cast<double>(v)),
                                                                                     std::fabs(static_cast<double>(k))))));
  }
  out.J += 0.5 * (((QN[0] * ey) * ey + (QN[1] * ep) * ep) + (QN[2] * t) * t);
  return out;
}

// One Riccati factorisation for the z-update's weights: stage i >= 1 and the terminal state carry rho_ey / rho_t on
// (e_y, t), every input R + rho_u.  Returns false on a singular step.
inline bool factor(const double* table, int n, const double Q[3], const double R[2], const double QN[3],
                   const double rho[4], Workspace& ws) {
  const double* kappa = table + 3 * static_cast<size_t>(n);
  const double* ds = table + 4 * static_cast<size_t>(n);
  const double* vel = table + 6 * static_cast<size_t>(n);
  ws.rows.resize(static_cast<size_t>(n) * 5);
  ws.fac.resize(static_cast<size_t>(n) * 18);
  ws.ks.resize(static_cast<size_t>(n) * 2);
  ws.zx.assign(static_cast<size_t>(n) * 2, 0.0);
  ws.zu.assign(static_cast<size_t>(n) * 2, 0.0);
  ws.trial.resize(static_cast<size_t>(n) * 2);
  const double R0 = R[0] + rho[2], R1 = R[1] + rho[3];
  double P00 = QN[0] + rho[0], P01 = 0.0, P02 = 0.0, P11 = QN[1], P12 = 0.0, P22 = QN[2] + rho[1];
  for (int i = n - 1; i >= 0; --i) {
    const double d = ds[i];
    const double a = -(kappa[i] * kappa[i]) * d;
    const double g = -kappa[i] / (vel[i] * d + lq::kEps);
    const double b = -1.0 / (vel[i] * vel[i] * d + lq::kEps);
    const double c = 1.0 / (vel[i] * d + lq::kEps);
    double* row = ws.rows.data() + static_cast<size_t>(i) * 5;
    row[0] = d, row[1] = a, row[2] = g, row[3] = b, row[4] = c;
    const double h0 = b * P02, h1 = b * P12, h2 = b * P22;
    const double m0 = d * P01, m1 = d * P11, m2 = d * P12;
    const double Quu00 = R0 + b * h2;
    const double Quu01 = d * h1;
    const double Quu11 = R1 + d * m1;
    const double S00 = (h0 + a * h1) + g * h2, S01 = d * h0 + h1, S02 = h2;
    const double S10 = (m0 + a * m1) + g * m2, S11 = d * m0 + m1, S12 = m2;
    const double det = Quu00 * Quu11 - Quu01 * Quu01;
    if (!(det > 0.0) || !std::isfinite(det)) return false;
    const double inv = 1.0 / det;
    const double I00 = inv * Quu11, I01 = -(inv * Quu01), I11 = inv * Quu00;
    const double K00 = -(I00 * S00 + I01 * S10), K01 = -(I00 * S01 + I01 * S11), K02 = -(I00 * S02 + I01 * S12);
    const double K10 = -(I01 * S00 + I11 * S10), K11 = -(I01 * S01 + I11 * S11), K12 = -(I01 * S02 + I11 * S12);
    double* F = ws.fac.data() + static_cast<size_t>(i) * 18;
    F[0] = K00, F[1] = K01, F[2] = K02, F[3] = K10, F[4] = K11, F[5] = K12;
    F[6] = I00, F[7] = I01, F[8] = I11;
    F[9] = S00, F[10] = S01, F[11] = S02, F[12] = S10, F[13] = S11, F[14] = S12;
    F[15] = P02, F[16] = P12, F[17] = P22;
    const double t00 = (P00 + a * P01) + g * P02, t10 = (P01 + a * P11) + g * P12, t20 = (P02 + a * P12) + g * P22;
    const double t01 = d * P00 + P01, t11 = d * P01 + P11, t21 = d * P02 + P12;
    const double t02 = P02, t12 = P12, t22 = P22;
    const double N00 = (t00 + a * t10) + g * t20, N01 = (t01 + a * t11) + g * t21, N02 = (t02 + a * t12) + g * t22;
    const double N11 = d * t01 + t11, N12 = d * t02 + t12, N22 = t22;
    const double q0 = i >= 1 ? Q[0] + rho[0] : Q[0], q2 = i >= 1 ? Q[2] + rho[1] : Q[2];
    const double n00 = (q0 + N00) + (S00 * K00 + S10 * K10);
    const double n01 = N01 + (S00 * K01 + S10 * K11);
    const double n02 = N02 + (S00 * K02 + S10 * K12);
    const double n11 = (Q[1] + N11) + (S01 * K01 + S11 * K11);
    const double n12 = N12 + (S01 * K02 + S11 * K12);
    const double n22 = (q2 + N22) + (S02 * K02 + S12 * K12);
    P00 = n00, P01 = n01, P02 = n02, P11 = n11, P12 = n12, P22 = n22;
  }
  return std::isfinite(P00) && std::isfinite(P11) && std::isfinite(P22);
}

inline double clip(double x, double lo, double hi) { return std::fmin(std::fmax(x, lo), hi); }

// `iterations` splitting iterations (at most) from `st` (cold when st.n != n), stopping early on the per-class
// tolerances; returns the iterations run.  On return ws.zu / st.wu hold the two iterates' inputs (du about u_ref).
inline int iterate(const double* table, int n, const double x0[3], const float u_lo[2], const float u_hi[2], double margin,
                   const double rho[4], int iterations, State& st, Workspace& ws) {
  const double* kappa = table + 3 * static_cast<size_t>(n);
  const double* width = table + 5 * static_cast<size_t>(n);
  const double* vel = table + 6 * static_cast<size_t>(n);
  const double lo_v = u_lo[0], lo_k = u_lo[1], hi_v = u_hi[0], hi_k = u_hi[1];
  if (st.n != n) {   // cold: w = the projection of "on the reference" into the box, no dual
    st.wx.assign(static_cast<size_t>(n) * 2, 0.0);
    st.wu.assign(static_cast<size_t>(n) * 2, 0.0);
    st.lx.assign(static_cast<size_t>(n) * 2, 0.0);
    st.lu.assign(static_cast<size_t>(n) * 2, 0.0);
    for (int i = 0; i < n; ++i) {
      const double half = width[i] / 2.0 - margin;
      st.wx[2 * i] = clip(0.0, -half, half);
      st.wx[2 * i + 1] = std::fmax(0.0, kTMin);
      st.wu[2 * i] = clip(0.0, lo_v - vel[i], hi_v - vel[i]);
      st.wu[2 * i + 1] = clip(0.0, lo_k - kappa[i], hi_k - kappa[i]);
    }
    st.n = n;
  }
  int it = 0;
  while (it < iterations) {
    ++it;
    // backward vector pass: p_i of the value function's linear term, k_i of the policy du_i = K_i x_i + k_i
    double p0 = -(rho[0] * (st.wx[2 * (n - 1)] - st.lx[2 * (n - 1)]));
    double p1 = 0.0;
    double p2 = -(rho[1] * (st.wx[2 * (n - 1) + 1] - st.lx[2 * (n - 1) + 1]));
    for (int i = n - 1; i >= 0; --i) {
      const double* row = ws.rows.data() + static_cast<size_t>(i) * 5;
      const double* F = ws.fac.data() + static_cast<size_t>(i) * 18;
      const double d = row[0], a = row[1], g = row[2], b = row[3], c = row[4];
      const double w0 = c * F[15] + p0, w1 = c * F[16] + p1, w2 = c * F[17] + p2;
      const double qu0 = -(rho[2] * (st.wu[2 * i] - st.lu[2 * i])) + b * w2;
      const double qu1 = -(rho[3] * (st.wu[2 * i + 1] - st.lu[2 * i + 1])) + d * w1;
      const double k0 = -(F[6] * qu0 + F[7] * qu1), k1 = -(F[7] * qu0 + F[8] * qu1);
      ws.ks[2 * i] = k0, ws.ks[2 * i + 1] = k1;
      double q0 = 0.0, q2 = 0.0;
      if (i >= 1) {
        q0 = -(rho[0] * (st.wx[2 * (i - 1)] - st.lx[2 * (i - 1)]));
        q2 = -(rho[1] * (st.wx[2 * (i - 1) + 1] - st.lx[2 * (i - 1) + 1]));
      }
      const double n0 = (q0 + ((w0 + a * w1) + g * w2)) + (F[9] * k0 + F[12] * k1);
      const double n1 = (d * w0 + w1) + (F[10] * k0 + F[13] * k1);
      const double n2 = (q2 + w2) + (F[11] * k0 + F[14] * k1);
      p0 = n0, p1 = n1, p2 = n2;
    }
    // forward rollout of the policy (the z iterate), then the box projection and the dual step, component by component
    double ey = x0[0], ep = x0[1], t = x0[2];
    double gap = 0.0, move = 0.0, drift = 0.0;   // largest tolerance-scaled |z - w|, |w+ - w| and |z+ - z|
    bool finite = true;
    for (int i = 0; i < n; ++i) {
      const double* row = ws.rows.data() + static_cast<size_t>(i) * 5;
      const double* F = ws.fac.data() + static_cast<size_t>(i) * 18;
      const double d = row[0], a = row[1], g = row[2], b = row[3], c = row[4];
      const double dv = ((F[0] * ey + F[1] * ep) + F[2] * t) + ws.ks[2 * i];
      const double dk = ((F[3] * ey + F[4] * ep) + F[5] * t) + ws.ks[2 * i + 1];
      const double ey_n = ey + d * ep;
      const double ep_n = (ep + a * ey) + d * dk;
      const double t_n = ((t + g * ey) + b * dv) + c;
      ey = ey_n, ep = ep_n, t = t_n;
      drift = std::fmax(drift, std::fmax(std::fmax(std::fabs(ey - ws.zx[2 * i]) * kPerTolEy, std::fabs(t - ws.zx[2 * i + 1]) * kPerTolT),
                                        std::fmax(std::fabs(dv - ws.zu[2 * i]) * kPerTolV, std::fabs(dk - ws.zu[2 * i + 1]) * kPerTolK)));
      ws.zu[2 * i] = dv, ws.zu[2 * i + 1] = dk;
      ws.zx[2 * i] = ey, ws.zx[2 * i + 1] = t;
      const double half = width[i] / 2.0 - margin;
      // box projection and dual step of the four boxed components of this step: z -> (w, l); `gap` / `move` in units of
      // the class's tolerance
      auto project = [&](double z, double lo, double hi, double per_tol, double& w, double& l) {
        const double relaxed = kAlpha * z + (1.0 - kAlpha) * w;
        const double next = clip(relaxed + l, lo, hi);
        l = (l + relaxed) - next;
        finite = finite && std::isfinite(relaxed) && std::isfinite(l);
        gap = std::fmax(gap, std::fabs(z - next) * per_tol);
        move = std::fmax(move, std::fabs(next - w) * per_tol);
        w = next;
      };
      project(ey, -half, half, kPerTolEy, st.wx[2 * i], st.lx[2 * i]);
      project(t, kTMin, HUGE_VAL, kPerTolT, st.wx[2 * i + 1], st.lx[2 * i + 1]);
      project(dv, lo_v - vel[i], hi_v - vel[i], kPerTolV, st.wu[2 * i], st.lu[2 * i]);
      project(dk, lo_k - kappa[i], hi_k - kappa[i], kPerTolK, st.wu[2 * i + 1], st.lu[2 * i + 1]);
    }
    if (!finite) {   // a NaN or an infinity somewhere: no iterate to keep
      st.reset();
      return -it;
    }
    // converged: the two iterates agree and the box iterate has stopped moving; or - box rows that NO plan can meet (the
    // reference pins t_0 = 0 while boxing t >= 0.01, control.py:134 vs :67: at 84 m/s on 3 m steps t_1 = 0.004) - both
    // iterates have stopped moving a gap apart: nothing more to gain either (from the second iteration on: the first has
    // no previous dynamics iterate to compare with)
    if (move <= 1.0 && (gap <= 1.0 || (it >= 2 && drift <= 1.0))) break;
  }
  return it;
}

struct Result { int iterations = 0; int chosen = 0; Cost cost; bool triggered = false; };   // chosen: 0 LQ, 1 w, 2 z

// The plan of the last sampling round's candidate 2 under acmpc_params::lq_candidate = 2.  `plan` holds the LQ plan of
// acmpc_lq.h on entry (lq::plan: true) and the cheapest of {LQ plan, w iterate, clipped z iterate} on return.  The
// splitting only runs when the LQ plan is not already the QP's optimum - a control on the input box, or state rows
// violated beyond the solver's own acceptance test (eps_abs + eps_rel |z|_inf per row, as sampling_solver.py applies it
// to the winner); otherwise the iterate is dropped (the next active tick starts cold).
inline Result refine(const double* table, int n, const double x0[3], const double Q[3], const double R[2], const double QN[3],
                     const float u_lo[2], const float u_hi[2], double margin, double w_bound, int iterations, State& st,
                     Workspace& ws, float* plan) {
  Result res;
  res.cost = rollout_cost(table, n, x0, Q, R, QN, u_lo, u_hi, margin, plan);
  const double accept = 1.0e-3 + 1.0e-3 * res.cost.biggest;
  res.triggered = res.cost.saturated || res.cost.V > accept * accept || !(res.cost.V == res.cost.V);
  if (!res.triggered || iterations < 1) {
    st.reset();
    return res;
  }
  const double rho[4] = {kRhoEy, kRhoT, R[0], R[1]};
  if (!(rho[2] > 0.0) || !(rho[3] > 0.0) || !factor(table, n, Q, R, QN, rho, ws)) {
    st.reset();
    return res;
  }
  // a warm iterate continues for at most kWarmIterations per call: what fits behind the prologue and the first round of a
  // tick (12 x 1.8 us + the factorisation and the rollouts, ~28 us: closed loop round a 9 m corner, same box, 8 / 12 / 16 / 40 per call:
  // tick p50 52.2 / 52.5 / 56.1 / 81 us; solves lost in an 8.2 m corner 218 / 179 / 171 / 160 of 1 790); one that needs more takes it over the next ticks
  const int budget = (st.n == n && iterations > kWarmIterations) ? kWarmIterations : iterations;
  res.iterations = iterate(table, n, x0, u_lo, u_hi, margin, rho, budget, st, ws);
  if (res.iterations < 0) return res;
  const double* kappa = table + 3 * static_cast<size_t>(n);
  const double* vel = table + 6 * static_cast<size_t>(n);
  double best = res.cost.J + w_bound * res.cost.V;
  for (int which = 1; which <= 2; ++which) {
    const double* du = which == 1 ? st.wu.data() : ws.zu.data();
    bool finite = true;
    for (int i = 0; i < n; ++i) {
      const float v = std::fmin(std::fmax(static_cast<float>(vel[i] + du[2 * i]), u_lo[0]), u_hi[0]);
      const float k = std::fmin(std::fmax(static_cast<float>(kappa[i] + du[2 * i + 1]), u_lo[1]), u_hi[1]);
      ws.trial[2 * i] = v, ws.trial[2 * i + 1] = k;
      finite = finite && std::isfinite(v) && std::isfinite(k);
    }
    if (!finite) continue;
    const Cost trial = rollout_cost(table, n, x0, Q, R, QN, u_lo, u_hi, margin, ws.trial.data());
    const double total = trial.J + w_bound * trial.V;
    if (total < best) {
      best = total;
      res.cost = trial;
      res.chosen = which;
      for (int i = 0; i < 2 * n; ++i) plan[i] = ws.trial[i];
    }
  }
  return res;
}

}  // namespace lqbox
}  // namespace acmpc
