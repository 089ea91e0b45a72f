// Host-side float64 arithmetic round one MPC solve, moved out of NumPy because at ~50 elements per array every
// NumPy call costs more in dispatch than in arithmetic (the closed-loop solve spent 45 of its 160 us there):
//
//   acmpc_waypoint_table    SpatialMPC.construct_waypoints          src/acmpc/control/spatial_mpc.py:125-154
//   acmpc_velocity_ceiling  SpeedProfileSolver's v_max vector       src/acmpc/control/solvers/speed_profile.py:26-43,131-150
//   acmpc_unpack_decision   the tail of SpatialMPC.get_control       src/acmpc/control/spatial_mpc.py:156-168,195-211
//
// Same operations in the same order as the NumPy statements in acmpc_amd/mpc.py and speed_profile.py (kept there for
// the tests, which hold the two to 1e-12 and pin both to the reference's golden vectors).  Plain C++: no GPU work.
#include <cmath>

#include "../../include/acmpc.h"

namespace {

const double kPi = 3.14159265358979323846;

// np.mod(a + pi, 2 pi) - pi: floored modulo, result in [-pi, pi)
inline double wrap_angle(double a) {
  const double two_pi = 2.0 * kPi;
  double m = std::fmod(a + kPi, two_pi);
  if (m != 0.0 && (m < 0.0)) m += two_pi;
  return m - kPi;
}

}  // namespace

extern "C" int acmpc_waypoint_table(const double* coords, int32_t H, double eps, double* table) {
  if (coords == nullptr || table == nullptr || H < 3) return ACMPC_EINVAL;
  const int n = H - 1;
  double* x = table;
  double* y = table + n;
  double* psi = table + 2 * n;
  double* kappa = table + 3 * n;
  double* ds = table + 4 * n;
  double* width = table + 5 * n;
  double* v = table + 6 * n;
  for (int i = 0; i < n; ++i) {
    const double* here = coords + 3 * i;
    const double* next = coords + 3 * (i + 1);
    const double* prev = coords + 3 * ((i == 0) ? H - 1 : i - 1);  // point 0 closes the loop to the last point
    const double ax = next[0] - here[0], ay = next[1] - here[1];
    const double bx = here[0] - prev[0], by = here[1] - prev[1];
    x[i] = here[0];
    y[i] = here[1];
    psi[i] = std::atan2(ay, ax);
    ds[i] = std::sqrt(ax * ax + ay * ay);
    width[i] = next[2];
    const double turn = wrap_angle(psi[i] - std::atan2(by, bx));
    kappa[i] = turn / (ds[i] + eps) + eps;
    v[i] = 0.0;
  }
  kappa[0] = kappa[1];
  return ACMPC_OK;
}

extern "C" int acmpc_velocity_ceiling(const double* kappa, int32_t n, double ay_max, double ki_min, double v_min,
                                      double v_max, int32_t localised, int32_t has_end_velocity, double end_velocity,
                                      double* ceiling) {
  if (kappa == nullptr || ceiling == nullptr || n < 1) return ACMPC_EINVAL;
  if (localised != 0) {  // the map-derived reference speed itself, no end velocity (speed_profile.py:131-150)
    for (int i = 0; i < n; ++i) ceiling[i] = v_max;
    return ACMPC_OK;
  }
  const double eps = 1e-12;
  for (int i = 0; i < n; ++i) {
    const double curvature = std::fabs(kappa[i]);
    double c = std::sqrt(ay_max / (curvature + eps));
    if (curvature < ki_min) c = v_max;
    c = std::fmin(c, v_max);
    c = std::fmax(v_min, c);
    ceiling[i] = c + 2.0;
  }
  if (has_end_velocity != 0) ceiling[n - 1] = end_velocity;
  return ACMPC_OK;
}

extern "C" int acmpc_unpack_decision(const double* z, int32_t n, const double* table, double wheelbase,
                                     double* projected_control, double* prediction, double* cum_time, double* times,
                                     double* accelerations, double* steer_rates) {
  if (z == nullptr || table == nullptr || projected_control == nullptr || prediction == nullptr ||
      cum_time == nullptr || times == nullptr || accelerations == nullptr || steer_rates == nullptr || n < 2)
    return ACMPC_EINVAL;
  // z = [x_0 .. x_n (3 each) ; u_0 .. u_{n-1} (2 each)]  (control.py:121-158)
  const double* states = z;                 // the first n of the n + 1 states are published (spatial_mpc.py:202)
  const double* controls = z + 3 * (n + 1);
  const double* xs = table;
  const double* ys = table + n;
  const double* psis = table + 2 * n;
  for (int i = 0; i < n; ++i) {
    projected_control[i] = controls[2 * i];                                   // v
    projected_control[n + i] = std::atan(controls[2 * i + 1] * wheelbase);    // delta = atan(kappa L)
    const double e_y = states[3 * i];
    prediction[2 * i] = xs[i] - e_y * std::sin(psis[i]);                      // s2t, dynamics.py:42-63
    prediction[2 * i + 1] = ys[i] + e_y * std::cos(psis[i]);
    cum_time[i] = states[3 * i + 2];
  }
  for (int i = 0; i + 1 < n; ++i) {
    times[i] = states[3 * (i + 1) + 2] - states[3 * i + 2];
    accelerations[i] = (states[3 * (i + 1)] - states[3 * i]) / times[i];
    steer_rates[i] = (states[3 * (i + 1) + 1] - states[3 * i + 1]) / times[i];
  }
  return ACMPC_OK;
}

// Mode T's counterpart: the record's states are the poses (X, Y, phi) after each Euler step of `dt` seconds, so the
// prediction is read off them, the time of step i is i dt, and - where the reference differentiates its Frenet states
// (spatial_mpc.py:209-211) - the plan's own controls are differentiated: accelerations = dv / dt, steer_rates =
// d(delta) / dt (build-defined: the reference has no temporal rollout).
extern "C" int acmpc_unpack_decision_temporal(const double* z, int32_t n, double dt, double wheelbase,
                                              double* projected_control, double* prediction, double* cum_time,
                                              double* times, double* accelerations, double* steer_rates) {
  if (z == nullptr || projected_control == nullptr || prediction == nullptr || cum_time == nullptr || times == nullptr ||
      accelerations == nullptr || steer_rates == nullptr || n < 2 || !(dt > 0.0))
    return ACMPC_EINVAL;
  const double* states = z;
  const double* controls = z + 3 * (n + 1);
  for (int i = 0; i < n; ++i) {
    projected_control[i] = controls[2 * i];
    projected_control[n + i] = std::atan(controls[2 * i + 1] * wheelbase);
    prediction[2 * i] = states[3 * i];
    prediction[2 * i + 1] = states[3 * i + 1];
    cum_time[i] = static_cast<double>(i) * dt;
  }
  for (int i = 0; i + 1 < n; ++i) {
    times[i] = dt;
    accelerations[i] = (projected_control[i + 1] - projected_control[i]) / dt;
    steer_rates[i] = (projected_control[n + i + 1] - projected_control[n + i]) / dt;
  }
  return ACMPC_OK;
}
