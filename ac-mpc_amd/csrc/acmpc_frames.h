// Frames of mode T's verified nearest-waypoint search: what the host (acmpc_set_paths -> verified_frames, acmpc_capi.hip)
// and the per-tick prologue kernel (acmpc_prologue.hip) both tabulate for the rollout kernels' acceptance test
// (acmpc_device.h: nearest_verified_window).  One definition of the arithmetic, compiled for both sides.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace acmpc {

// window of mode T's verified nearest-waypoint search (exhaustive semantics): waypoints searched per step and how many
// of them lie behind the previous step's nearest one (A/B builds: ACMPC_HIPCC_EXTRA="-DACMPC_VERIFIED_WINDOW=16 ...")
#ifndef ACMPC_VERIFIED_WINDOW
#define ACMPC_VERIFIED_WINDOW 8
#define ACMPC_VERIFIED_BACK 3
#endif
constexpr int kVerifiedWindow = ACMPC_VERIFIED_WINDOW;
constexpr int kVerifiedBack = ACMPC_VERIFIED_BACK;
// the table of one problem, per window position: [t_x, t_y, k_along, k_across | slab, tube, far, -slack] (two 16-byte reads)
constexpr int kFrameStride = 8;
constexpr float kFrameAcrossMax = 32.0f;     // [m] cap of `across`: bounds how far from the path a certified pose can be
constexpr float kFrameVirtualPlane = 32.0f;  // [m] where the plane of an EMPTY side (window at the path's end) is put
__host__ __device__ constexpr int verified_frame_floats(int n) { return kFrameStride * (n - kVerifiedWindow + 1); }

namespace frames {

// From the float32 waypoint positions the kernels use: for every window position lo (window = waypoints lo .. lo + W - 1) eight floats
// [t_x, t_y, k_along, k_across, slab, tube, far, -slack] with which the kernel evaluates, at a pose p,
//     alpha = t . p + k_along     (distance past the plane BEHIND which every near waypoint m < lo lies)
//     beta  = n . p + k_across    (n = (-t_y, t_x); offset from the middle of the tube that holds every near outside waypoint)
//     along = med3(alpha, slab - alpha, 0),   across = med3(|beta| - tube, 0, kFrameAcrossMax)
// and certifies the window's winner j when  |r_j^2 recovered from its key|  <  min(along^2 + across^2 - slack, far).
//
// The outside waypoints are split by index: NEAR = within `near` waypoints of the window's ends, FAR = the rest; `near`
// is the smallest of first, first + 8, ... for which the far ones are at least kFarReach from the window (all of them near
// when none is: a path that comes back to the window); first = 16, 32 or 64 by the path's spacing (near_first()).
//
// Near waypoints: why the slab-and-tube bound is sound.  t is the float32 direction of the window's chord scaled so that
// |t| <= 1, and n has the same norm, so for any waypoint w: (t.(w - p))^2 + (n.(w - p))^2 <= |w - p|^2.  Every near
// waypoint m < lo has t.w <= a_B, every near m >= lo + W has t.w >= a_A (a_B, a_A = the extreme projections, taken here
// over the same float32 vectors), and every near outside waypoint has |n.w - mid| <= tube; hence |w - p|^2 >= along^2 +
// across^2 in exact arithmetic.  A side without waypoints (window at an end of the path) gets a virtual plane
// kFrameVirtualPlane beyond the window: any plane is valid there, a finite one keeps `slab` - and with it the distance of
// a certified pose from the path - bounded.  Rounding: alpha and beta are two fused multiply-adds each; their error at a
// certified pose is below `delta` (computed per window from the magnitudes involved); k_along is lowered by delta, slab by
// 4 delta, tube raised by 3 delta, each rounded outward, so that the kernel's along / across never exceed the exact ones.
// What remains - the rounding of the two squares, twice the error E_key of a key and the error E_est of the recovered
// r_j^2 - is `slack`: a certified pose satisfies r_est < D := slab_max^2 + kFrameAcrossMax^2, so |p| <= Lc with Lc found
// below such that any pose further out has r_est >= D whatever its rounding; every intermediate of a key is then below
// B = 6 Lc^2 in magnitude, a key off by at most 3 B ulp (two roundings in c, two fused multiply-adds), the recovered r^2
// by at most 7 B ulp.  (Where keys overflow - poses beyond 1e17 m - the winning key is -inf and so is the recovered
// r^2: hence its magnitude in the test.)
//
// Far waypoints: with R = the smallest distance between a waypoint of the window and a far one, a pose within r of the
// winner is at least R - r from every far waypoint, whose squared distance therefore exceeds r^2 by R (R - 2 r);
//     far = ((R - 2 E_key / R) / (2 (1 + 1e-3)))^2 - E_est,   rounded down
// keeps that gap above twice a key's error (+inf when there is no far waypoint).
//
// A window whose chord has no length, or a path with a non-finite waypoint: slab = -1, tube = +inf, far = 0 - never
// certified.  O(n W) per window and choice of `near`, O(n^2) per path.
#ifndef ACMPC_FRAME_FAR_REACH
#define ACMPC_FRAME_FAR_REACH 45.0
#endif
constexpr double kFarReach = ACMPC_FRAME_FAR_REACH;   // [m]: poses up to half of this from the winner are not cut off by the far bound
constexpr double kUlp = 5.9604644775390625e-08;   // 2^-24: half an ulp of a float32 of magnitude 1, one rounding's relative error

struct Geometry {
  double tx, ty, aB, aA, mid, tube, R;
  bool usable;
};

// `at(m, x, y)` yields waypoint m's float32 position as doubles.
//
// The far split of the window lo .. lo + W - 1: distance from the window to the waypoints more than `near` indices beyond
// its ends (+inf when there is none), rounded towards the window.  squared_gap() is the one expression every caller
// takes a pair's squared distance from (the prologue kernel finds the same minimum in another order, acmpc_prologue.hip):
// float32 arithmetic on the float32 positions - three roundings, under 2e-7 relative - which far_from_squared() more
// than takes back (a distance that overflows float32 is no constraint on poses within the slab's reach of the path).
__host__ __device__ inline float squared_gap(float x_far, float y_far, float x_win, float y_win) {
  const float dx = x_far - x_win, dy = y_far - y_win;
  return dx * dx + dy * dy;
}
__host__ __device__ inline double far_from_squared(float R2) { return sqrt(static_cast<double>(R2)) * (1.0 - 1.0e-6); }

// The first choice of `near` (then + 8, + 16, ...): 16 waypoints at the spacing of a racing line's reference path (150 m in
// 50 points), 32 / 64 on denser paths (the mapping controller's 100 points) - the next ~45 m of path either side.  Decided
// by the path's largest gap between neighbours, taken with squared_gap(): a maximum, the same whatever order it is taken in.
__host__ __device__ inline int near_first(float largest_gap2) {
  return (largest_gap2 >= 2.5f * 2.5f) ? 16 : (largest_gap2 >= 1.25f * 1.25f) ? 32 : 64;
}

template <typename At>
__host__ __device__ inline double far_distance(const At& at, int n, int lo, int near) {
  const int hi = lo + kVerifiedWindow - 1;
  float R2 = __builtin_inff();
  for (int m = 0; m < n; ++m) {
    if (m >= lo - near && m <= hi + near) {
      m = hi + near;   // (skip the window and its near neighbourhood)
      continue;
    }
    double xm, ym;
    at(m, xm, ym);
    for (int q = lo; q <= hi; ++q) {
      double xq, yq;
      at(q, xq, yq);
      const float d2 = squared_gap(static_cast<float>(xm), static_cast<float>(ym), static_cast<float>(xq), static_cast<float>(yq));
      R2 = (d2 < R2) ? d2 : R2;
    }
  }
  return far_from_squared(R2);
}

// `near` = the smallest of first, first + 8, ... that leaves the far waypoints at least kFarReach from the window.  On
// entry R is the far distance for near = first; on return for the `near` chosen.
template <typename At>
__host__ __device__ inline int choose_near(const At& at, int n, int lo, int first, double& R) {
  int near = first;
  while (R < kFarReach) {
    near += 8;
    R = far_distance(at, n, lo, near);   // (+inf once nothing is far)
  }
  return near;
}

// The window's geometry for a split (near, R) as choose_near() leaves it.  NEAR > 0: `near` is that compile-time value
// (the loop over the neighbourhood unrolls and its loads go out together - the prologue kernel's single wavefront waits
// for every LDS round trip it cannot overlap); NEAR = 0: any `near`.  The same operations on the same operands.
template <int NEAR, typename At>
__host__ __device__ inline Geometry window_geometry(const At& at, int n, int lo, bool finite, int near_any, double R) {
  constexpr int W = kVerifiedWindow;
  const int near = (NEAR > 0) ? NEAR : near_any;
  const double inf = __builtin_huge_val();
  Geometry f{};
  f.usable = false;
  f.R = R;
  const int hi = lo + W - 1;
  double xl, yl, xh, yh;
  at(lo, xl, yl);
  at(hi, xh, yh);
  const double cx = xh - xl, cy = yh - yl, chord = sqrt(cx * cx + cy * cy);
  if (!finite || !(chord > 0.0) || !(chord < inf)) return f;
  // float32 direction with |t| <= 1: shrunk by more than its two roundings can add
  f.tx = static_cast<double>(static_cast<float>(cx / chord * (1.0 - 4.0e-7)));
  f.ty = static_cast<double>(static_cast<float>(cy / chord * (1.0 - 4.0e-7)));
  double aB = -inf, aA = inf, lowest = inf, highest = -inf, first = inf, last = -inf;
  bool behind = false, ahead = false;
  auto take = [&](int k) {   // waypoint lo + k of the neighbourhood, when the path has it
    const int m = lo + k;
    const bool there = m >= 0 && m < n;
    double xm, ym;
    at(there ? m : lo, xm, ym);
    const double a = f.tx * xm + f.ty * ym, b = -f.ty * xm + f.tx * ym;
    // (every value here is finite - the caller's `finite` - so fmin / fmax are plain minima and maxima: one instruction)
    if (there && k < 0) {
      aB = fmax(aB, a);
      behind = true;
    }
    if (there && k >= W) {
      aA = fmin(aA, a);
      ahead = true;
    }
    if (there && (k < 0 || k >= W)) {
      lowest = fmin(lowest, b);
      highest = fmax(highest, b);
    }
    if (k >= 0 && k < W) {
      first = fmin(first, a);
      last = fmax(last, a);
    }
  };
  if constexpr (NEAR > 0) {
#pragma unroll
    for (int k = -NEAR; k < W + NEAR; ++k) take(k);
  } else {
    for (int k = -near; k < W + near; ++k) take(k);
  }
  if (!behind) aB = first - static_cast<double>(kFrameVirtualPlane);
  if (!ahead) aA = last + static_cast<double>(kFrameVirtualPlane);
  if (!behind && !ahead) lowest = highest = 0.0;   // no near outside waypoint at all: any tube will do
  f.aB = aB;
  f.aA = aA;
  f.mid = 0.5 * (lowest + highest);
  f.tube = 0.5 * (highest - lowest);
  f.usable = aA > aB;
  return f;
}

// What a path's windows share: how far out a certified pose can be, and the rounding errors that follow from it.
// wn = the largest norm of a waypoint, slab_max = the largest aA - aB of a usable window.
struct Scale {
  double Lc, e_key, e_est, slack;
  bool ok;
};

__host__ __device__ inline Scale path_scale(double wn, double slab_max) {
  // r_est < D for a certified pose, and beyond Lc the recovered r^2 is at least D whatever it rounds to
  const double across_max = static_cast<double>(kFrameAcrossMax);
  const double D = slab_max * slab_max + across_max * across_max;
  Scale s{};
  s.Lc = wn + 1.01 * sqrt(D) + 1.0;
  while (!((s.Lc - wn) * (s.Lc - wn) - 42.0 * kUlp * (s.Lc + wn) * (s.Lc + wn) >= 1.01 * D) && s.Lc < 1.0e12) s.Lc *= 1.5;
  const double B = 6.0 * s.Lc * s.Lc;
  s.e_key = 3.0 * B * kUlp;
  s.e_est = 7.0 * B * kUlp;
  s.slack = 1.01 * (2.0 * s.e_key + s.e_est + 8.0 * kUlp * D);
  s.ok = s.Lc < 1.0e12;
  return s;
}

// The neighbours of a finite float32 (the next representable value up / down, without a library call on the device).
__host__ __device__ inline float float_above(float v) {
  if (v == 0.0f) return 1.401298464324817e-45f;
  int bits = __builtin_bit_cast(int, v);
  bits += (v > 0.0f) ? 1 : -1;
  return __builtin_bit_cast(float, bits);
}
__host__ __device__ inline float float_below(float v) { return -float_above(-v); }

// One window's eight floats.
__host__ __device__ inline void frame_row(const Geometry& f, const Scale& s, float* row) {
  const float finf = __builtin_inff();
  row[0] = row[1] = row[2] = row[3] = 0.0f;
  row[4] = -1.0f;
  row[5] = finf;
  row[6] = 0.0f;
  row[7] = -float_above(static_cast<float>(s.slack));
  if (!f.usable || !s.ok) return;
  // |alpha|, |beta| evaluated at |X|, |Y| <= Lc: each of the two fused multiply-adds rounds a value below
  // 2 Lc + |k|; the subtraction that follows (slab - alpha, |beta| - tube) one below 2 Lc + |k| + slab + tube
  const double k_along = -f.aB, k_across = -f.mid;
  const double delta = kUlp * (8.0 * s.Lc + 4.0 * (fabs(k_along) + fabs(k_across)) + 2.0 * ((f.aA - f.aB) + f.tube));
  const double slab = (f.aA - f.aB) - 4.0 * delta;
  if (!(slab > 0.0)) return;
  float far = finf;
  if (f.R < __builtin_huge_val()) {
    const double rho = (f.R - 2.0 * s.e_key / f.R) / (2.0 * (1.0 + 1.0e-3));
    const double bound = rho * rho - s.e_est;
    if (!(f.R > 0.0) || !(rho > 0.0) || !(bound > 0.0) || !(bound < __builtin_huge_val())) return;
    far = float_below(static_cast<float>(bound));
  }
  row[0] = static_cast<float>(f.tx);
  row[1] = static_cast<float>(f.ty);
  row[2] = float_below(static_cast<float>(k_along - delta));
  row[3] = static_cast<float>(k_across);
  row[4] = float_below(static_cast<float>(slab));
  row[5] = float_above(static_cast<float>(f.tube + 3.0 * delta));
  row[6] = far;
}

}  // namespace frames
}  // namespace acmpc
