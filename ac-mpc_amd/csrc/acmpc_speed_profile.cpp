// Host-side solver of the reference's speed-profile QP (src/acmpc/control/solvers/speed_profile.py:26-59), which the
// reference hands to the third-party OSQP package: the team-of-one instantiation of acmpc_admm.h (the same statement
// of the algorithm the device prologue runs on a wavefront).  O(n) per iteration, which is what makes the whole-lap
// profile of the race start (n ~ 10^4 waypoints, spatial_mpc.py:60-87) as cheap per iteration as the 49-point horizon
// profile of every tick.  Plain C++: no GPU work.
#include <vector>

#include "../../include/acmpc.h"
#include "acmpc_admm.h"

extern "C" int acmpc_speed_profile_qp(const double* v_hi, const double* ds, int32_t n, double a_min, double a_max,
                                      double v_min, int32_t max_iter, int32_t check_every, double eps_abs,
                                      double eps_rel, double* v, double* y, int32_t warm_start, int32_t* iterations) {
  if (v_hi == nullptr || ds == nullptr || v == nullptr || y == nullptr || n < 2) return ACMPC_EINVAL;
  std::vector<double> workspace(static_cast<size_t>(acmpc::admm::workspace_doubles(n)));
  acmpc::admm::Workspace w;
  w.bind(workspace.data(), n);
  const acmpc::admm::Settings s{a_min, a_max, v_min, max_iter, check_every > 0 ? check_every : 10, eps_abs, eps_rel};
  int its = 0;
  const int status = acmpc::admm::solve(acmpc::admm::HostTeam{}, w, v_hi, ds, n, s, v, y, warm_start, &its);
  if (iterations != nullptr) *iterations = its;
  return status;
}

// The same QP's exact optimum in two passes (acmpc_admm.h: exact_profile) - what the tick's prologue tries first.
extern "C" int acmpc_speed_profile_exact(const double* v_hi, const double* ds, int32_t n, double a_min, double a_max,
                                         double v_min, double* v, double* y) {
  if (v_hi == nullptr || ds == nullptr || v == nullptr || y == nullptr || n < 2) return ACMPC_EINVAL;
  double red[8];
  acmpc::admm::Workspace w{};   // (only its reduction scratch is used)
  w.red = red;
  std::vector<double> scratch(static_cast<size_t>(4) * n);
  const acmpc::admm::Settings s{a_min, a_max, v_min, 0, 1, 0.0, 0.0};
  return acmpc::admm::exact_profile(acmpc::admm::HostTeam{}, w, v_hi, ds, n, s, v, y, scratch.data()) ? 0 : 1;
}
