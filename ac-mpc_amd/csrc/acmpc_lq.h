// The LQ plan: one deterministic candidate for the sampling rounds (round 4).
//
// The reference's controller solves a QP (control/solvers/control.py:26-79): quadratic cost on (x - x_ref, u - u_ref),
// the linearised spatial bicycle model as equality rows, and boxes on inputs and states.  WITHOUT the boxes that QP is a
// finite-horizon time-varying LQ problem with affine dynamics, whose optimum a backward Riccati pass gives exactly:
//     x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i,   J = sum_i 1/2 (x_i' Q x_i + du_i' R du_i) + 1/2 x_n' QN x_n
//     V_i(x) = 1/2 x' P_i x + p_i' x + const,   du_i = K_i x_i + k_i
// The plan rolls that feedback forward from the start state and clips every control into the QP's input box as it goes
// (the feedback then sees the state the clipped controls really produce): where no box is active it IS the QP optimum,
// where the steering saturates it stays within a per cent of it (28 scenarios of the reference's own MPC script: excess
// over the optimum's tracking cost median 0.000, worst 0.011 - the sampled plans of round 3: median 0.25, worst 12).
// It enters the last sampling round as candidate 2 (SampleArgs::u_extra) and is rolled and costed like every other
// candidate: the argmin keeps it only when it wins.
//
// Host only, float64, one fixed operation order, no fused multiply-add (the library is built with -ffp-contract=off):
// oracle/acmpc_oracle.py lq_plan() restates it line by line and is bit-identical.  A_i, B_i, f_i are those of
// SpatialBicycleModel.linearise (control/dynamics.py:65-103) from the 7 x n table:
//     A = [[1, ds, 0], [a, 1, 0], [g, 0, 1]],  B = [[0, 0], [0, ds], [b, 0]],  f = (0, 0, c)
//     a = -kappa^2 ds,  g = -kappa / (v ds + eps),  b = -1 / (v^2 ds + eps),  c = 1 / (v ds + eps),  u_ref = (v, kappa)
#pragma once
#include <cmath>
#include <vector>

namespace acmpc {
namespace lq {

constexpr double kEps = 1e-12;   // spatial_mpc.py:34, dynamics.py:21

// table: 7 x n float64, rows [x, y, psi, kappa, ds, width, v] (control/paths.py:4-72); x0 = (e_y, e_psi, t);
// u_lo / u_hi: the QP's input box as the kernels hold it (float32); plan: [n][2] float32 (v, kappa).
// Returns false when a step's 2 x 2 system is singular or anything turns non-finite (no plan: the round runs without).
inline bool plan(const double* table, int n, const double x0[3], const double Q[3], const double R[2], const double QN[3],
                 const float u_lo[2], const float u_hi[2], float* out) {
  if (n < 1) return false;
  const double* kappa = table + 3 * static_cast<size_t>(n);
  const double* ds = table + 4 * static_cast<size_t>(n);
  const double* vel = table + 6 * static_cast<size_t>(n);
  std::vector<double> gains(static_cast<size_t>(n) * 8);   // K00 K01 K02 k0 | K10 K11 K12 k1 per step
  double P00 = QN[0], P01 = 0.0, P02 = 0.0, P11 = QN[1], P12 = 0.0, P22 = QN[2];
  double p0 = 0.0, p1 = 0.0, p2 = 0.0;
  for (int i = n - 1; i >= 0; --i) {
    const double d = ds[i];
    const double a = -(kappa[i] * kappa[i]) * d;
    const double g = -kappa[i] / (vel[i] * d + kEps);
    const double b = -1.0 / (vel[i] * vel[i] * d + kEps);
    const double c = 1.0 / (vel[i] * d + kEps);
    // P B: first column b (P02, P12, P22), second column d (P01, P11, P12)
    const double h0 = b * P02, h1 = b * P12, h2 = b * P22;
    const double m0 = d * P01, m1 = d * P11, m2 = d * P12;
    const double Quu00 = R[0] + b * h2;
    const double Quu01 = d * h1;
    const double Quu11 = R[1] + d * m1;
    // S = B' P A (2 x 3)
    const double S00 = (h0 + a * h1) + g * h2, S01 = d * h0 + h1, S02 = h2;
    const double S10 = (m0 + a * m1) + g * m2, S11 = d * m0 + m1, S12 = m2;
    // w = P f + p,  qu = B' w
    const double w0 = c * P02 + p0, w1 = c * P12 + p1, w2 = c * P22 + p2;
    const double qu0 = b * w2, qu1 = d * w1;
    const double det = Quu00 * Quu11 - Quu01 * Quu01;
    if (!(det > 0.0) || !std::isfinite(det)) return false;
    const double inv = 1.0 / det;
    const double K00 = -inv * (Quu11 * S00 - Quu01 * S10), K01 = -inv * (Quu11 * S01 - Quu01 * S11),
                 K02 = -inv * (Quu11 * S02 - Quu01 * S12);
    const double K10 = -inv * (Quu00 * S10 - Quu01 * S00), K11 = -inv * (Quu00 * S11 - Quu01 * S01),
                 K12 = -inv * (Quu00 * S12 - Quu01 * S02);
    const double k0 = -inv * (Quu11 * qu0 - Quu01 * qu1), k1 = -inv * (Quu00 * qu1 - Quu01 * qu0);
    double* G = gains.data() + static_cast<size_t>(i) * 8;
    G[0] = K00, G[1] = K01, G[2] = K02, G[3] = k0, G[4] = K10, G[5] = K11, G[6] = K12, G[7] = k1;
    // T = P A, N = A' T (upper triangle)
    const double t00 = (P00 + a * P01) + g * P02, t10 = (P01 + a * P11) + g * P12, t20 = (P02 + a * P12) + g * P22;
    const double t01 = d * P00 + P01, t11 = d * P01 + P11, t21 = d * P02 + P12;
    const double t02 = P02, t12 = P12, t22 = P22;
    const double N00 = (t00 + a * t10) + g * t20, N01 = (t01 + a * t11) + g * t21, N02 = (t02 + a * t12) + g * t22;
    const double N11 = d * t01 + t11, N12 = d * t02 + t12, N22 = t22;
    // P <- Q + A' P A + S' K,   p <- A' w + S' k
    const double n00 = (Q[0] + N00) + (S00 * K00 + S10 * K10);
    const double n01 = N01 + (S00 * K01 + S10 * K11);
    const double n02 = N02 + (S00 * K02 + S10 * K12);
    const double n11 = (Q[1] + N11) + (S01 * K01 + S11 * K11);
    const double n12 = N12 + (S01 * K02 + S11 * K12);
    const double n22 = (Q[2] + N22) + (S02 * K02 + S12 * K12);
    const double q0 = ((w0 + a * w1) + g * w2) + (S00 * k0 + S10 * k1);
    const double q1 = (d * w0 + w1) + (S01 * k0 + S11 * k1);
    const double q2 = w2 + (S02 * k0 + S12 * k1);
    P00 = n00, P01 = n01, P02 = n02, P11 = n11, P12 = n12, P22 = n22;
    p0 = q0, p1 = q1, p2 = q2;
  }
  double ey = x0[0], ep = x0[1], t = x0[2];
  bool finite = true;
  for (int i = 0; i < n; ++i) {
    const double d = ds[i];
    const double a = -(kappa[i] * kappa[i]) * d;
    const double g = -kappa[i] / (vel[i] * d + kEps);
    const double b = -1.0 / (vel[i] * vel[i] * d + kEps);
    const double c = 1.0 / (vel[i] * d + kEps);
    const double* G = gains.data() + static_cast<size_t>(i) * 8;
    const double dv = ((G[0] * ey + G[1] * ep) + G[2] * t) + G[3];
    const double dk = ((G[4] * ey + G[5] * ep) + G[6] * t) + G[7];
    // one rounding to float32, then the clip the sampler applies to every candidate (blend_control: fmin(fmax(u, lo), hi))
    const float v = std::fmin(std::fmax(static_cast<float>(vel[i] + dv), u_lo[0]), u_hi[0]);
    const float k = std::fmin(std::fmax(static_cast<float>(kappa[i] + dk), u_lo[1]), u_hi[1]);
    out[2 * i] = v;
    out[2 * i + 1] = k;
    finite = finite && std::isfinite(v) && std::isfinite(k);
    const double cv = static_cast<double>(v) - vel[i], ck = static_cast<double>(k) - kappa[i];   // what the clip left of du
    const double ey_n = ey + d * ep;
    const double ep_n = (ep + a * ey) + d * ck;
    const double t_n = ((t + g * ey) + b * cv) + c;
    ey = ey_n, ep = ep_n, t = t_n;
  }
  return finite;
}

// Frenet start state of a pose w.r.t. the path's first waypoint (SpatialBicycleModel.t2s, dynamics.py:23-40): what mode T
// handles - whose rollouts start from the pose itself - hand to plan()
inline void frenet_start(const double* table, int n, const double pose[3], double x0[3]) {
  const double xr = table[0], yr = table[static_cast<size_t>(n)], psir = table[2 * static_cast<size_t>(n)];
  x0[0] = std::cos(psir) * (pose[1] - yr) - std::sin(psir) * (pose[0] - xr);
  const double two_pi = 2.0 * M_PI;
  double wrapped = std::fmod(pose[2] - psir + M_PI, two_pi);
  if (wrapped < 0.0) wrapped += two_pi;   // np.mod's sign convention
  x0[1] = wrapped - M_PI;
  x0[2] = 0.0;
}

}  // namespace lq
}  // namespace acmpc
