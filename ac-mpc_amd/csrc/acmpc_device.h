// Device-side arithmetic of the rollout-and-cost path, gfx950 only.
//
// Every function here is the float32 "spec order" of DESIGN.md: one fixed association, no library transcendentals,
// and no IMPLICIT fused multiply-add (the translation unit is built with -ffp-contract=off and the pragma below).
// Mode S uses no FMA at all.  Mode T's specification names its FMAs explicitly (fma_() below = IEEE fmaf, one
// rounding): the oracle restates them with C's fmaf and, in NumPy, with an exact emulation (float64 product +
// TwoSum + round-to-odd).  That is what makes costs bit-identical to oracle/acmpc_oracle.{py,c}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "acmpc_frames.h"

#pragma clang fp contract(off)

namespace acmpc {

constexpr int kWave = 64;

__device__ __forceinline__ int wave_min_int(int v);
__device__ __forceinline__ int wave_sum_int(int v);
constexpr int kCoefS = 12;  // ACMPC_COEF_STRIDE_SPATIAL
constexpr int kCoefT = 8;   // ACMPC_COEF_STRIDE_TEMPORAL

// Scalars of the cost and the bounds; passed by value in the kernel argument segment, so they live in SGPRs.
struct Weights {
  float q0, q1, q2;     // Q   (control.py:126)
  float r0, r1;         // R   (control.py:127)
  float qn0, qn1, qn2;  // QN  (control.py:128)
  // mode T accumulates J += (w/2 * a) * a term by term: the halved weights 0.5f * w (float32 products, exact)
  float hq0, hq1, hr0, hr1, hqn0, hqn1, hqn2;
  float ulo0, ulo1, uhi0, uhi1;  // input box incl. the 0.1 m/s slack (control.py:130-139)
  float tmin;           // 0.01 (control.py:134)
  float wbound;
  float dt;
  int nn_back, nn_ahead;  // mode T search window round the previous nearest index; nn_ahead < 0 = exhaustive
};

// The step functions are written once for F = float (one candidate per lane) and F = f32x2 (two candidates per
// lane).  With two candidates in an ext_vector the adds/multiplies become v_pk_add_f32 / v_pk_mul_f32 - on gfx950 a
// wave64 VALU instruction issues every 4 cycles per SIMD whether it carries one or two floats per lane, so packing
// halves the issue slots of everything that has a packed form.  Per element the operations and their order are
// identical, so results do not depend on F.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma_(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float abs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ f32x2 abs_(f32x2 a) { return __builtin_elementwise_abs(a); }

template <typename F>
struct IndexOf {
  using type = int;
};
template <>
struct IndexOf<f32x2> {
  using type = i32x2;
};

template <typename F>
__device__ __forceinline__ F splat(float v) {
  return F(v);
}
template <typename F>
__device__ __forceinline__ F vmax(F a, F b) {
  return __builtin_elementwise_max(a, b);
}

template <typename F>
__device__ __forceinline__ F quad(float w, F a) {
  return (w * a) * a;
}

// fused multiply-add with ONE rounding (v_fma_f32 / v_fmac_f32 / v_pk_fma_f32): only where the specification says so
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename F>
__device__ __forceinline__ F hinge2(F lo_minus_x, F x_minus_hi) {
  // at most one side of a (non-degenerate) interval can be violated: one v_max3_f32
  const F v = vmax(vmax(lo_minus_x, x_minus_hi), splat<F>(0.0f));
  return v * v;
}

// ---- mode S --------------------------------------------------------------------------------------------
// One step of x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i (dynamics.py:65-103, control.py:26-45) with the
// stage cost 1/2 (x'Qx + du'R du) (control.py:72-79,151-158) and the squared violation of the input box, the
// corridor of x_{i+1} (control.py:57-60) and t >= t_min (control.py:134).
template <typename F>
struct StateS_ {
  F ey, ep, t, J, V;
};
using StateS = StateS_<float>;

template <typename F>
__device__ __forceinline__ void step_spatial(StateS_<F>& s, const float* __restrict__ c, F v, F k, const Weights& w) {
  const F dv = v - c[5];
  const F dk = k - c[6];
  F a = quad(w.q0, s.ey);
  a = a + quad(w.q1, s.ep);
  a = a + quad(w.q2, s.t);
  F r = quad(w.r0, dv);
  r = r + quad(w.r1, dk);
  s.J = s.J + 0.5f * (a + r);
  s.V = s.V + hinge2<F>(w.ulo0 - v, v - w.uhi0);
  s.V = s.V + hinge2<F>(w.ulo1 - k, k - w.uhi1);
  const float ds = c[0];
  const F ey_n = s.ey + ds * s.ep;
  const F ep_n = (s.ep + c[1] * s.ey) + ds * dk;
  const F t_n = ((s.t + c[2] * s.ey) + c[3] * dv) + c[4];
  s.ey = ey_n;
  s.ep = ep_n;
  s.t = t_n;
  s.V = s.V + hinge2<F>(c[7] - s.ey, s.ey - c[8]);
  const F tv = vmax(w.tmin - s.t, splat<F>(0.0f));
  s.V = s.V + tv * tv;
}

// step_spatial() in two halves that share nothing but the state recurrence: J never reads V and V never reads J, so a
// small launch - CUs to spare, one candidate's step a serial stream of ~50 instructions on a lone wave - can roll every
// candidate on TWO waves, one accumulating the stage cost and one the bound violations, each running the 9-operation
// recurrence itself (no hand-over inside the loop).  Every operation below is the corresponding line of step_spatial on
// the same operands in the same order: s.J of the first and s.V of the second are step_spatial's, bit for bit.
__device__ __forceinline__ void advance_spatial(StateS& s, const float* __restrict__ c, float dv, float dk) {
  const float ds = c[0];
  const float ey_n = s.ey + ds * s.ep;
  const float ep_n = (s.ep + c[1] * s.ey) + ds * dk;
  const float t_n = ((s.t + c[2] * s.ey) + c[3] * dv) + c[4];
  s.ey = ey_n;
  s.ep = ep_n;
  s.t = t_n;
}

__device__ __forceinline__ void step_spatial_cost(StateS& s, const float* __restrict__ c, float v, float k,
                                                  const Weights& w) {
  const float dv = v - c[5];
  const float dk = k - c[6];
  float a = quad(w.q0, s.ey);
  a = a + quad(w.q1, s.ep);
  a = a + quad(w.q2, s.t);
  float r = quad(w.r0, dv);
  r = r + quad(w.r1, dk);
  s.J = s.J + 0.5f * (a + r);
  advance_spatial(s, c, dv, dk);
}

__device__ __forceinline__ void step_spatial_bounds(StateS& s, const float* __restrict__ c, float v, float k,
                                                    const Weights& w) {
  s.V = s.V + hinge2<float>(w.ulo0 - v, v - w.uhi0);
  s.V = s.V + hinge2<float>(w.ulo1 - k, k - w.uhi1);
  advance_spatial(s, c, v - c[5], k - c[6]);
  s.V = s.V + hinge2<float>(c[7] - s.ey, s.ey - c[8]);
  const float tv = vmax(w.tmin - s.t, 0.0f);
  s.V = s.V + tv * tv;
}

template <typename F>
__device__ __forceinline__ F finish_spatial(const StateS_<F>& s, const Weights& w) {
  F a = quad(w.qn0, s.ey);
  a = a + quad(w.qn1, s.ep);
  a = a + quad(w.qn2, s.t);
  const F J = s.J + 0.5f * a;
  return J + w.wbound * s.V;
}

// ---- mode T --------------------------------------------------------------------------------------------
// sin / cos of the heading: Cody-Waite reduction by pi (phi = k pi + r, |r| <= pi/2, two-term pi) and odd / even
// polynomials on [-pi/2, pi/2] fitted for this build (Lawson-weighted least squares, coefficients rounded to
// float32; evaluated in float32 with the FMAs below |error| < 1.5e-7), every multiply-add fused: the same operation
// sequence as oracle sincos_spec().  t = fma(phi, 1/pi, 1.5 * 2^23) holds k = rint(phi / pi) in its low mantissa
// bits - no float -> int conversion (defined for every input, the same bits on every platform) - and
//     sin phi = (-1)^k sin r,   cos phi = (-1)^k cos r:
// a reduction by pi needs no quadrant swap, only the sign bit `bits(t) << 31`, at the price of two more polynomial
// terms than a reduction by pi/2 (the rollout is bound by instruction issue, and a select costs twice an FMA).
constexpr float kInvPi = 0.3183098861837907f;
constexpr float kPiHi = 3.140625f;            // 9 significant bits: k * kPiHi is exact for |k| < 2^15
constexpr float kPiLo = 9.67653589793e-4f;    // pi - kPiHi
constexpr float kSinC[4] = {-0.16666656732559204f, 0.008333016186952591f, -0.00019806546333711594f,
                            2.59990065387683e-06f};
constexpr float kCosC[5] = {-0.5f, 0.04166664183139801f, -0.0013888402609154582f, 2.4761806344031356e-05f,
                            -2.607563374112942e-07f};

// |sin|-side and |cos|-side values s = sin r, c = cos r and the sign mask (0 or 0x80000000) of (-1)^k
template <typename F>
__device__ __forceinline__ void sincos_reduced(F phi, F& s, F& c, typename IndexOf<F>::type& sign) {
  using I = typename IndexOf<F>::type;
  const F t = fma_(phi, splat<F>(kInvPi), splat<F>(12582912.0f));
  const F k = t - 12582912.0f;
  F r = fma_(-k, splat<F>(kPiHi), phi);
  r = fma_(-k, splat<F>(kPiLo), r);
  const F r2 = r * r;
  F ps = fma_(r2, splat<F>(kSinC[3]), splat<F>(kSinC[2]));
  ps = fma_(r2, ps, splat<F>(kSinC[1]));
  ps = fma_(r2, ps, splat<F>(kSinC[0]));
  s = fma_(r * r2, ps, r);
  F pc = fma_(r2, splat<F>(kCosC[4]), splat<F>(kCosC[3]));
  pc = fma_(r2, pc, splat<F>(kCosC[2]));
  pc = fma_(r2, pc, splat<F>(kCosC[1]));
  pc = fma_(r2, pc, splat<F>(kCosC[0]));
  c = fma_(r2, pc, splat<F>(1.0f));
  sign = __builtin_bit_cast(I, t) << 31;
}

template <typename F>
__device__ __forceinline__ void sincos_spec(F phi, F& sn, F& cs) {
  using I = typename IndexOf<F>::type;
  F s, c;
  I sign;
  sincos_reduced<F>(phi, s, c, sign);
  sn = __builtin_bit_cast(F, __builtin_bit_cast(I, s) ^ sign);
  cs = __builtin_bit_cast(F, __builtin_bit_cast(I, c) ^ sign);
}

// angle difference into [-pi, pi]: a - 2 pi rint(a / (2 pi)), the integer read off the magic-number sum like the k of
// sincos_reduced (three instructions; defined for every input; round 3 - the floor form took five)
template <typename F>
__device__ __forceinline__ F wrap_spec(F a) {
  const F t = fma_(a, splat<F>(0.159154943091895f), splat<F>(12582912.0f));
  const F q = t - 12582912.0f;
  return fma_(-q, splat<F>(6.28318530717959f), a);
}

// S0..S3: the running sums of e_y^2, e_psi^2, dv^2, dkappa^2 - the stage cost's weights are applied once, after the
// horizon (finish_temporal); V: the summed squared bound violations
template <typename F>
struct StateT_ {
  F X, Y, phi, ey, ep, S0, S1, S2, S3, V;
};
using StateT = StateT_<float>;

// Mode T works in the PATH'S OWN FRAME (round 4): every position - the pose, the waypoints, the search keys, the frames of
// the verified search - is taken relative to the path's first waypoint (x_0, y_0), subtracted in float32 when the tables
// are staged and when the pose is read, and added back where a pose leaves a kernel (records, traces).  Dynamics and cost
// only see differences of positions, so nothing else changes; but the search key |p - w|^2 - |p|^2 = c + a X + b Y cancels
// catastrophically in float32 when |p| is large: given in a frame 4 km from its origin, a path made the key pick a
// non-nearest waypoint for 5 % of the poses within 8 m of it (up to 1.2 m farther than the nearest).  In the path's own
// frame the coordinates are the path's extent (<= ~200 m) wherever the caller put it.  `coef` = the problem's packed table,
// whose first row holds (x_0, y_0).  For a path that starts at the origin - every path in the vehicle frame whose first
// point is the car's - x - 0 = x: the same bits as before.
template <typename F>
__device__ __forceinline__ StateT_<F> start_temporal(const float* __restrict__ pose, const float* __restrict__ coef) {
  const F zero = splat<F>(0.0f);
  return StateT_<F>{splat<F>(pose[0] - coef[0]), splat<F>(pose[1] - coef[1]), splat<F>(pose[2]), zero, zero, zero, zero, zero,
                    zero, zero};
}

// explicit Euler on the rear-axle kinematic bicycle (localiser.py:66-95); phi_dot = v * kappa
template <typename F>
__device__ __forceinline__ void temporal_advance(StateT_<F>& s, F v, F k, const Weights& w) {
  // specification: X += (v cos phi) dt with cos phi = (-1)^k cos r.  The sign is applied to v once instead of to the
  // sine and the cosine: (-v) c and v (-c) are the same float32 product
  using I = typename IndexOf<F>::type;
  F sr, cr;
  I sign;
  sincos_reduced<F>(s.phi, sr, cr, sign);
  const F vs = __builtin_bit_cast(F, __builtin_bit_cast(I, v) ^ sign);
  const F Xn = fma_(vs * cr, splat<F>(w.dt), s.X);
  const F Yn = fma_(vs * sr, splat<F>(w.dt), s.Y);
  const F phin = fma_(v * k, splat<F>(w.dt), s.phi);
  s.X = Xn;
  s.Y = Yn;
  s.phi = phin;
}

// ---- nearest waypoint ------------------------------------------------------------------------------------------
// Specification (round 3): the nearest waypoint is the FIRST minimum, under `<`, of the search key
//     e_m(p) = fma(Y, b_m, fma(X, a_m, c_m)),   a_m = -2 x_m,  b_m = -2 y_m,  c_m = fma(y_m, y_m, x_m x_m)
// - the squared distance |p - w_m|^2 less the term |p|^2 that is the same for every waypoint, two fused multiply-adds per
// waypoint where (X - x)^2 + (Y - y)^2 takes four operations (the rollout is bound by instruction issue, and the search
// was a third of its instructions).  a, b, c are float32, computed from the float32 waypoint positions once per table
// (a and b exactly; c with two roundings), so oracle/acmpc_oracle.{py,c} restate the key bit for bit.  In the vehicle
// frame (|coordinates| <= ~200 m) the key resolves ~0.01 m^2: it moves the boundary between two neighbouring
// waypoints by a millimetre.  NaN keys never win; when every key is NaN the answer is the first waypoint searched.
template <typename F, typename G>
__device__ __forceinline__ F search_key(F X, F Y, G a, G b, G c) {
  return fma_(Y, F(b), fma_(X, F(a), F(c)));
}
// The key table in LDS, one 32-byte entry per waypoint m (round 4):
//     [a_m, b_m, a_m+1, b_m+1 | c_m, c_m+1, c_m+2, c_m+3]
// - the entry of waypoint m repeats what its successors' entries hold, so that a search window that starts at ANY
// waypoint reads its keys with 16-byte loads from 16-byte aligned addresses: an 8-waypoint window is four ds_read_b128
// of (a, b) pairs (entries lo, lo + 2, lo + 4, lo + 6) and two of c (entries lo, lo + 4) - six LDS instructions of four
// LDS-array cycles each (MI355X_MICROARCH.md, LDS table: ds_read_b128 moves 256 B per clock, ds_read2_b32 128) where
// the (a, b, c)-side-by-side table of round 3 took twelve ds_read2_b32 of four cycles.  The mode T rollout kernel keeps
// the CU's LDS array busy for 70 % of its run time (profiles/r03_mode_T_sq_counters.json); the keys were three quarters of it.
// Entries past the path's end repeat the last waypoint (never searched: a window ends at waypoint n - 1).
// Loops that visit one waypoint at a time read a_m, b_m, c_m at kKeyStride * m + 0, kKeyB, kKeyC.
// (Measured before, round 3, on the side-by-side table and not kept: padding (a, b, c) to 16 bytes, 161 us against
// 138-152; (a, b) pairs + planar c through ds_read2_b64 / ds_read2_b32 - the same LDS cycles as before: a wash.)
constexpr int kKeyStride = 8;
constexpr int kKeyB = 1;
constexpr int kKeyC = 4;

// the key's table entries for one waypoint (x, y): a, b, c
__device__ __forceinline__ void search_entry(float x, float y, float& a, float& b, float& c) {
  a = -2.0f * x;
  b = -2.0f * y;
  c = fma_(y, y, x * x);
}
// |p - w_m|^2 back from the key (to ~0.01 m^2): what the verified search tests against its frame's bound
__device__ __forceinline__ float distance2_of_key(float X, float Y, float e) { return fma_(Y, Y, fma_(X, X, e)); }

template <typename F, typename G>
__device__ __forceinline__ F dist2(F X, F Y, G wx, G wy) {
  const F dx = X - wx;
  const F dy = Y - wy;
  return fma_(dy, dy, dx * dx);
}

// every lane scans the whole table (wave-uniform addresses: the LDS read is a broadcast and serves both candidates of a
// packed lane).  `abc` = the key table, (a, b, c) per waypoint side by side (stage_temporal_tables).
template <typename F>
__device__ __forceinline__ typename IndexOf<F>::type temporal_nearest(const StateT_<F>& s, const float* abc, int n) {
  using I = typename IndexOf<F>::type;
  F best = splat<F>(__builtin_inff());
  I j = I(0);
  int i = 0;
  for (; i + 4 <= n; i += 4) {   // four waypoints per trip: three 16-byte reads (entry i: a, b of i, i + 1 and c of i .. i + 3)
    const f32x4* e = reinterpret_cast<const f32x4*>(abc + kKeyStride * i);
    const f32x4 ab0 = e[0], c = e[1], ab1 = e[4];
    const float a[4] = {ab0[0], ab0[2], ab1[0], ab1[2]}, b[4] = {ab0[1], ab0[3], ab1[1], ab1[3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const F d = search_key<F>(s.X, s.Y, a[q], b[q], c[q]);
      const auto better = d < best;
      best = better ? d : best;
      j = better ? I(i + q) : j;
    }
  }
  for (; i < n; ++i) {
    const F d = search_key<F>(s.X, s.Y, abc[kKeyStride * i], abc[kKeyStride * i + kKeyB], abc[kKeyStride * i + kKeyC]);
    const auto better = d < best;
    best = better ? d : best;
    j = better ? I(i) : j;
  }
  return j;
}

// The keys of the W consecutive waypoints from `lo` (W = 4, 8, 16): W / 2 + W / 4 ds_read_b128 from one address (see
// kKeyStride), then two fused multiply-adds per waypoint.
template <int W>
__device__ __forceinline__ void window_keys(float X, float Y, const float* abc, int lo, float (&d)[W]) {
  static_assert(W % 4 == 0, "whole 16-byte groups of c");
  const f32x4* first = reinterpret_cast<const f32x4*>(abc + kKeyStride * lo);   // (16-byte aligned: 32-byte entries)
  f32x4 ab[W / 2], c[W / 4];
#pragma unroll
  for (int m = 0; m < W / 2; ++m) ab[m] = first[4 * m];        // entry lo + 2m: a, b of waypoints lo + 2m, lo + 2m + 1
#pragma unroll
  for (int m = 0; m < W / 4; ++m) c[m] = first[8 * m + 1];     // entry lo + 4m: c of waypoints lo + 4m .. lo + 4m + 3
#pragma unroll
  for (int m = 0; m < W; ++m)
    d[m] = search_key<float>(X, Y, ab[m / 2][2 * (m & 1)], ab[m / 2][2 * (m & 1) + 1], c[m / 4][m & 3]);
}

// The same search restricted to W = back + ahead + 1 consecutive waypoints starting at
// lo = clamp(j_prev - back, 0, n - W): progress along the path is monotone and at most about one waypoint per step,
// so a short window finds the global minimum on every realistic input at a fraction of the ALU work (tests check
// equality with the exhaustive scan).  W = 16, 8 and 4 are unrolled: the minimum by a min chain, and the FIRST
// index that attains it by an equality scan - the same answer as the `d < best` scan, cheaper.
template <int W>
__device__ __forceinline__ int nearest_in_window(float X, float Y, const float* abc, int lo, float* best_out = nullptr) {
  float d[W];
  window_keys<W>(X, Y, abc, lo, d);
  float best = d[0];
#pragma unroll
  for (int m = 1; m < W; ++m) best = __builtin_fminf(best, d[m]);  // NaN keys are skipped, like `d < best`
  // FIRST index that attains the minimum (0 when nothing compares equal - all NaN - where the `d < best` scan keeps
  // `lo`).  The W compare results are lane masks in scalar registers; which window position holds each lane's first
  // hit is worked out on the scalar unit (prefix ORs, one-hot first hits, one mask per index bit), so the vector unit
  // sees W compares and log2(W) selects instead of W compares and W selects.
  static_assert((W & (W - 1)) == 0 && W <= 16, "window widths with an unrolled search");
  unsigned long long hit[W], before = 0ull, bit[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
  for (int m = 0; m < W; ++m) hit[m] = __builtin_amdgcn_ballot_w64(d[m] == best);
#pragma unroll
  for (int m = 0; m < W; ++m) {
    const unsigned long long first = hit[m] & ~before;  // lanes whose first hit is position m
    before |= hit[m];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((m >> q) & 1) bit[q] |= first;
  }
  int jm = __builtin_amdgcn_inverse_ballot_w64(bit[0]) ? 1 : 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if ((1 << q) < W) jm |= __builtin_amdgcn_inverse_ballot_w64(bit[q]) ? (1 << q) : 0;
  if (best_out != nullptr) *best_out = best;
  return lo + jm;
}

// EXHAUSTIVE semantics at windowed cost.  The kVerifiedWindow waypoints from lo = clamp(j_prev - kVerifiedBack, 0,
// n - kVerifiedWindow) are searched as above, and a certificate decides whether the window's first minimum IS the global
// first minimum.  For every window position lo the host tabulates a frame (acmpc_capi.hip: verified_frames) that splits
// the waypoints OUTSIDE the window in two:
//   - the NEAR ones (the next few dozen metres of path either side) lie behind the window or ahead of it and, seen from
//     the window, inside a tube round the path.  The frame holds a direction t along the window's chord (|t| <= 1) and
//     its normal, the plane behind which every near earlier waypoint lies, the plane beyond which every near later one
//     lies, and the tube's half-width across.  With
//         along  = distance of the pose to the nearer of the two planes (0 outside the slab between them),
//         across = what the pose's lateral offset exceeds the tube by (0 inside it, capped at kFrameAcrossMax),
//     every near outside waypoint is at least sqrt(along^2 + across^2) away - a bound that does not decay when a
//     candidate runs wide of the path, which is where the candidates of a sampling round are (a ball round the winning
//     waypoint, round 2's certificate, needs twice the window for the same yield);
//   - the FAR ones are at least D from every waypoint of the window, so further than r from a pose within r < D / 2 of
//     the window's winner.
// The winner j is certified when the squared distance recovered from its key is below min(along^2 + across^2 - slack,
// far) - slack and the margin inside `far` keep the gap above twice a key's rounding error plus the error of the
// recovered distance and of the frame arithmetic.  Otherwise (a path that folds back on itself, a non-finite position)
// all waypoints are scanned - by the whole wave, see nearest_cooperative_fix().  Either way the index is exactly the
// exhaustive one, bit for bit.
// (kVerifiedWindow, kVerifiedBack, the frame's layout and the host / prologue side of the table: acmpc_frames.h)

// The frame test for the window that starts at `lo`: is the first minimum `best` (a key) of that window the global one?
__device__ __forceinline__ bool frame_certifies(float X, float Y, float best, const float* frames, int lo) {
  const f32x4* f = reinterpret_cast<const f32x4*>(frames) + (kFrameStride / 4) * lo;   // (16-byte aligned)
  const f32x4 t = f[0], ext = f[1];
  const float alpha = fma_(t[0], X, fma_(t[1], Y, t[2]));    // along the chord, from the plane behind the window
  const float beta = fma_(t[0], Y, fma_(-t[1], X, t[3]));    // across, from the tube's middle
  const float along = __builtin_amdgcn_fmed3f(alpha, ext[0] - alpha, 0.0f);
  const float across = __builtin_amdgcn_fmed3f(__builtin_fabsf(beta) - ext[1], 0.0f, kFrameAcrossMax);
  const float bound = __builtin_fminf(fma_(across, across, fma_(along, along, ext[3])), ext[2]);
  // (the magnitude: a pose so far out that its keys overflow to -inf must not pass - and costs nothing, a source modifier)
#ifdef ACMPC_DEBUG_ALWAYS_CERTIFIED   // (timing experiment only: what the kernel costs without its fallback; WRONG results)
  return true;
#endif
  return __builtin_fabsf(distance2_of_key(X, Y, best)) < bound;
}

// The window part: index of the window's first minimum and whether the frame test certifies it as global.
__device__ __forceinline__ int nearest_verified_window(float X, float Y, const float* abc, const float* frames, int n,
                                                       int j_prev, bool& certified) {
  const int lo = max(min(j_prev - kVerifiedBack, n - kVerifiedWindow), 0);
  float best;
  const int j = nearest_in_window<kVerifiedWindow>(X, Y, abc, lo, &best);
  certified = frame_certifies(X, Y, best, frames, lo);
  return j;
}

// The fallback, wave-cooperative: for every lane whose window result is not certified (a handful per wave and
// step), ALL 64 lanes scan the key table side by side for that lane's position and reduce to the first minimum
// instead of one lane walking all n waypoints while the other 63 wait.  Must be reached by every lane of the wave (the
// rollout kernel runs its tail lanes on a valid dummy candidate for this reason).  Same answer as the one-lane scan:
// NaN keys never win, ties go to the lower index, nothing finite found -> 0.
__device__ __forceinline__ int nearest_cooperative_fix(float X, float Y, bool certified, int j, const float* abc, int n) {
  unsigned long long pending = __ballot(!certified);
  const int lane = static_cast<int>(__lane_id());
  constexpr int kNothing = 0x7f800000;   // the ordered pattern of +inf: what a lane without a finite key holds
  while (pending != 0ull) {
    const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(pending)) - 1);
    pending &= pending - 1ull;
    const float px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(X), src));
    const float py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Y), src));
    int found;
    if (n <= kWave) {
      // one waypoint per lane: the smallest key (keys order as their sign-flipped bit patterns), then the first lane
      // that holds it - a ballot instead of a second reduction
      const int m = min(lane, n - 1);
      const float d = search_key<float>(px, py, abc[kKeyStride * m], abc[kKeyStride * m + kKeyB], abc[kKeyStride * m + kKeyC]);
      const int bits = __float_as_int(d);
      const int ordered = (lane < n && d < __builtin_inff()) ? ((bits >= 0) ? bits : (bits ^ 0x7fffffff)) : kNothing;
      const int least = wave_min_int(ordered);
      const unsigned long long holders = __ballot(ordered == least);
      found = (least == kNothing) ? 0 : __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(holders)) - 1);
    } else {
      float best = __builtin_inff();
      int jj = 0x7fffffff;
      for (int m = lane; m < n; m += kWave) {
        const float d = search_key<float>(px, py, abc[kKeyStride * m], abc[kKeyStride * m + kKeyB], abc[kKeyStride * m + kKeyC]);
        const bool better = d < best;
        best = better ? d : best;
        jj = better ? m : jj;
      }
      // first minimum over the lanes: the smallest key (+inf where a lane found nothing), then the lowest index among
      // the lanes that hold it
      const int bits = __float_as_int(best);
      const int ordered = (bits >= 0) ? bits : (bits ^ 0x7fffffff);
      const int least = wave_min_int(ordered);
      found = wave_min_int((ordered == least) ? jj : 0x7fffffff);
      found = (found == 0x7fffffff) ? 0 : found;
    }
    j = (lane == src) ? found : j;
  }
  return j;
}

__device__ __forceinline__ int nearest_verified(float X, float Y, const float* abc, const float* frames, int n, int j_prev) {
  bool certified;
  const int j = nearest_verified_window(X, Y, abc, frames, n, j_prev, certified);
  return nearest_cooperative_fix(X, Y, certified, j, abc, n);
}

// How a kernel searches: decided once per launch (the weights are wave-uniform), so the step loop itself is
// branch-free and the scheduler can interleave the searches of the candidates a lane owns.
constexpr int kSearchExhaustive = 0;
constexpr int kSearchGeneric = -1;   // any window width, rolled loop; 8 and 4 name the unrolled widths
constexpr int kSearchVerified = -2;  // exhaustive semantics through nearest_verified() (needs the frame table)

__device__ __forceinline__ int search_kind(const Weights& w, int n, bool has_frames = false) {
  if (w.nn_ahead < 0) return (has_frames && n >= kVerifiedWindow) ? kSearchVerified : kSearchExhaustive;
  const int W = w.nn_back + w.nn_ahead + 1;
  return ((W == 8 || W == 4) && n >= W) ? W : kSearchGeneric;
}

template <int SEARCH>
__device__ __forceinline__ int temporal_nearest_window(float X, float Y, const float* abc, int n, int j_prev, int back,
                                                       int ahead) {
  const int W = (SEARCH > 0) ? SEARCH : back + ahead + 1;
  const int lo = max(min(j_prev - back, n - W), 0);
  if constexpr (SEARCH > 0) {
    return nearest_in_window<SEARCH>(X, Y, abc, lo);
  } else {
    const int hi = min(lo + W, n);
    float best = __builtin_inff();
    int j = lo;
    for (int i = lo; i < hi; ++i) {
      const float d = search_key<float>(X, Y, abc[kKeyStride * i], abc[kKeyStride * i + kKeyB], abc[kKeyStride * i + kKeyC]);
      const bool better = d < best;
      best = better ? d : best;
      j = better ? i : j;
    }
    return j;
  }
}

// The waypoint rows as the kernels keep them in LDS, derived from the table's [x, y, cos psi, sin psi, psi, k_ref, v_ref,
// w/2 - margin] once per workgroup: [s x - c y, -s, c, psi, k_ref, v_ref, w/2 - margin, 0], so that the lateral error
// e_y = c (Y - y) - s (X - x) (dynamics.py:23-40) is two fused multiply-adds on the pose; and the search key's entries
// (search_entry) in the 32-byte entries described at kKeyStride.  `threads` lanes of a workgroup cooperate.
__device__ __forceinline__ void stage_temporal_tables(const float* __restrict__ coef, int n, int tid, int threads,
                                                      float* rows, float* abc) {
  const float ox = coef[0], oy = coef[1];   // the path's own frame: see start_temporal()
  for (int m = tid; m < n; m += threads) {
    const float* g = coef + m * kCoefT;
    const float x = g[0] - ox, y = g[1] - oy, c = g[2], sn = g[3];
    float* r = rows + m * kCoefT;
    r[0] = fma_(sn, x, -(c * y));
    r[1] = -sn;
    r[2] = c;
    r[3] = g[4];
    r[4] = g[5];
    r[5] = g[6];
    r[6] = g[7];
    r[7] = 0.0f;
    float* e = abc + kKeyStride * m;
    float unused_a, unused_b;
    search_entry(x, y, e[0], e[1], e[kKeyC]);
    const float* g1 = coef + min(m + 1, n - 1) * kCoefT;
    search_entry(g1[0] - ox, g1[1] - oy, e[2], e[3], e[kKeyC + 1]);
    const float* g2 = coef + min(m + 2, n - 1) * kCoefT;
    search_entry(g2[0] - ox, g2[1] - oy, unused_a, unused_b, e[kKeyC + 2]);
    const float* g3 = coef + min(m + 3, n - 1) * kCoefT;
    search_entry(g3[0] - ox, g3[1] - oy, unused_a, unused_b, e[kKeyC + 3]);
  }
}

__device__ __forceinline__ float med3_(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ f32x2 med3_(f32x2 x, f32x2 lo, f32x2 hi) {
  f32x2 out;
  out[0] = __builtin_amdgcn_fmed3f(x[0], lo[0], hi[0]);
  out[1] = __builtin_amdgcn_fmed3f(x[1], lo[1], hi[1]);
  return out;
}

// Frenet errors w.r.t. the (derived) waypoint row(s) g, the stage cost's squares and the bound violations.  The excess
// over the input box is x - med3(x, lo, hi): the same magnitude as max(lo - x, x - hi, 0), one instruction less.
template <typename F>
__device__ __forceinline__ void temporal_cost(StateT_<F>& s, const F (&g)[kCoefT], F v, F k, const Weights& w) {
  s.ey = fma_(g[2], s.Y, fma_(g[1], s.X, g[0]));
  s.ep = wrap_spec<F>(s.phi - g[3]);
  const F dv = v - g[5];
  const F dk = k - g[4];
  s.S0 = fma_(s.ey, s.ey, s.S0);
  s.S1 = fma_(s.ep, s.ep, s.S1);
  s.S2 = fma_(dv, dv, s.S2);
  s.S3 = fma_(dk, dk, s.S3);
  const F hv = v - med3_(v, splat<F>(w.ulo0), splat<F>(w.uhi0));
  s.V = fma_(hv, hv, s.V);
  const F hk = k - med3_(k, splat<F>(w.ulo1), splat<F>(w.uhi1));
  s.V = fma_(hk, hk, s.V);
  const F hc = vmax(abs_(s.ey) - g[6], splat<F>(0.0f));  // outside the corridor |e_y| <= w/2 - margin
  s.V = fma_(hc, hc, s.V);
}

// A derived waypoint row: two 16-byte LDS reads.  (Left to itself the compiler reads the second half - whose last float is
// padding - with ds_read_b96: eight LDS-array cycles where ds_read_b128 takes four.  The empty asm makes all four lanes of
// the vector live.)
__device__ __forceinline__ void load_row(const float* wp, int j, float (&row)[kCoefT]) {
  const f32x4* r = reinterpret_cast<const f32x4*>(wp + j * kCoefT);
  f32x4 lo = r[0], hi = r[1];
  asm volatile("" : "+v"(hi));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    row[q] = lo[q];
    row[4 + q] = hi[q];
  }
}

// one-candidate form used by the finalize kernel and the tile kernel
__device__ __forceinline__ void temporal_cost(StateT& s, const float* g, float v, float k, const Weights& w) {
  float row[kCoefT];
  load_row(g, 0, row);
  temporal_cost<float>(s, row, v, k, w);
}

// gather the table row(s) of the nearest waypoint(s) and account the step
__device__ __forceinline__ void temporal_settle(StateT& s, const float* wp, int j, float v, float k, const Weights& w) {
  temporal_cost(s, wp + j * kCoefT, v, k, w);
}
__device__ __forceinline__ void temporal_settle(StateT_<f32x2>& s, const float* wp, i32x2 j, f32x2 v, f32x2 k,
                                                const Weights& w) {
  f32x2 g[kCoefT];
  float g0[kCoefT], g1[kCoefT];
  load_row(wp, j[0], g0);
  load_row(wp, j[1], g1);
#pragma unroll
  for (int q = 0; q < kCoefT; ++q) {
    g[q][0] = g0[q];
    g[q][1] = g1[q];
  }
  temporal_cost<f32x2>(s, g, v, k, w);
}

// The verified search in two phases, so that a kernel can run the window phase of ALL the candidates a lane owns
// before the (rare, wave-wide) fallback phase and keep the window searches free of control flow.
__device__ __forceinline__ int verified_window(const StateT& s, const float* abc, const float* frames, int n, int j_prev,
                                               int& uncertified) {
  bool ok;
  const int j = nearest_verified_window(s.X, s.Y, abc, frames, n, j_prev, ok);
  uncertified = ok ? 0 : 1;
  return j;
}
__device__ __forceinline__ i32x2 verified_window(const StateT_<f32x2>& s, const float* abc, const float* frames, int n,
                                                 i32x2 j_prev, int& uncertified) {
  bool ok0, ok1;
  i32x2 j;
  j[0] = nearest_verified_window(s.X[0], s.Y[0], abc, frames, n, j_prev[0], ok0);
  j[1] = nearest_verified_window(s.X[1], s.Y[1], abc, frames, n, j_prev[1], ok1);
  uncertified = (ok0 ? 0 : 1) | (ok1 ? 0 : 2);
  return j;
}
__device__ __forceinline__ int verified_fix(const StateT& s, const float* abc, int n, int j, int uncertified) {
  return nearest_cooperative_fix(s.X, s.Y, uncertified == 0, j, abc, n);
}
__device__ __forceinline__ i32x2 verified_fix(const StateT_<f32x2>& s, const float* abc, int n, i32x2 j, int uncertified) {
  j[0] = nearest_cooperative_fix(s.X[0], s.Y[0], (uncertified & 1) == 0, j[0], abc, n);
  j[1] = nearest_cooperative_fix(s.X[1], s.Y[1], (uncertified & 2) == 0, j[1], abc, n);
  return j;
}

// `wp` holds the derived waypoint rows and `abc` the search keys' entries (stage_temporal_tables); in the rollout
// kernels both live in LDS.  Returns the nearest index (per element) for the next step's
// search window.
template <int SEARCH>
__device__ __forceinline__ int step_temporal_as(StateT& s, const float* wp, const float* abc, int n, float v, float k,
                                                const Weights& w, int j_prev, const float* frames = nullptr) {
  temporal_advance<float>(s, v, k, w);
  int j;
  if constexpr (SEARCH == kSearchExhaustive) {
    j = temporal_nearest<float>(s, abc, n);
  } else if constexpr (SEARCH == kSearchVerified) {
    j = nearest_verified(s.X, s.Y, abc, frames, n, j_prev);
  } else {
    j = temporal_nearest_window<SEARCH>(s.X, s.Y, abc, n, j_prev, w.nn_back, w.nn_ahead);
  }
  temporal_cost(s, wp + j * kCoefT, v, k, w);
  return j;
}

template <int SEARCH>
__device__ __forceinline__ i32x2 step_temporal_as(StateT_<f32x2>& s, const float* wp, const float* abc, int n, f32x2 v,
                                                  f32x2 k, const Weights& w, i32x2 j_prev, const float* frames = nullptr) {
  temporal_advance<f32x2>(s, v, k, w);
  i32x2 j;
  if constexpr (SEARCH == kSearchExhaustive) {
    j = temporal_nearest<f32x2>(s, abc, n);
  } else if constexpr (SEARCH == kSearchVerified) {
    j[0] = nearest_verified(s.X[0], s.Y[0], abc, frames, n, j_prev[0]);
    j[1] = nearest_verified(s.X[1], s.Y[1], abc, frames, n, j_prev[1]);
  } else {
    j[0] = temporal_nearest_window<SEARCH>(s.X[0], s.Y[0], abc, n, j_prev[0], w.nn_back, w.nn_ahead);
    j[1] = temporal_nearest_window<SEARCH>(s.X[1], s.Y[1], abc, n, j_prev[1], w.nn_back, w.nn_ahead);
  }
  temporal_settle(s, wp, j, v, k, w);
  return j;
}

// the search of one step alone (the three-wave round splits a step into pose / search / cost).  Run by a LONE wave: there
// every instruction costs the same ~2 ns, scalar ones included, so the unrolled windows use the plain `d < best` chain
// (three vector instructions per waypoint) instead of nearest_in_window's min chain + equality masks, whose ~30 scalar
// mask operations are free only where other waves fill the gaps.  The same first minimum either way.
template <int W>
__device__ __forceinline__ int nearest_in_window_chain(float X, float Y, const float* abc, int lo, float& best) {
  float d[W];
  window_keys<W>(X, Y, abc, lo, d);
  best = __builtin_inff();
  int j = lo;
#pragma unroll
  for (int m = 0; m < W; ++m) {
    const bool better = d[m] < best;
    best = better ? d[m] : best;
    j = better ? lo + m : j;
  }
  return j;
}

__device__ __forceinline__ int verified_window_start(int j_prev, int n) {
  return max(min(j_prev - kVerifiedBack, n - kVerifiedWindow), 0);
}

// The verified search in two halves, for kernels whose search runs on a wavefront of its own (the three-wave round): the
// searching wave only PROPOSES - the first minimum of the window that starts at verified_window_start(previous
// proposal) - and moves on; a wave with time to spare CONFIRMS: the proposal's key is recomputed (the same expression:
// the same bits), the frame of the window it came from decides, and what is not certified is scanned by the whole
// wave.  The result does not depend on where the window was - it is the nearest of all waypoints either way - so a
// proposal that turns out wrong only costs the following windows their good position, never the answer.
__device__ __forceinline__ int propose_nearest(float X, float Y, const float* abc, int n, int proposal_before) {
  float best;
  return nearest_in_window_chain<kVerifiedWindow>(X, Y, abc, verified_window_start(proposal_before, n), best);
}
__device__ __forceinline__ int confirm_nearest(float X, float Y, const float* abc, const float* frames, int n,
                                               int proposal_before, int proposal) {
  const float* e = abc + kKeyStride * proposal;
  const float best = search_key<float>(X, Y, e[0], e[kKeyB], e[kKeyC]);   // (a NaN key is below no bound: the scan decides, as it must)
  const int lo = verified_window_start(proposal_before, n);
  return nearest_cooperative_fix(X, Y, frame_certifies(X, Y, best, frames, lo), proposal, abc, n);
}

// (the verified search of a lone searching wave is propose_nearest() + confirm_nearest() above)
template <int SEARCH>
__device__ __forceinline__ int search_temporal_as(float X, float Y, const float* abc, int n, const Weights& w, int j_prev) {
  static_assert(SEARCH != kSearchVerified, "propose_nearest / confirm_nearest");
  if constexpr (SEARCH == kSearchExhaustive) {
    StateT probe{};
    probe.X = X;
    probe.Y = Y;
    return temporal_nearest<float>(probe, abc, n);
  } else if constexpr (SEARCH > 0) {
    const int lo = max(min(j_prev - w.nn_back, n - SEARCH), 0);
    float best;
    return nearest_in_window_chain<SEARCH>(X, Y, abc, lo, best);
  } else {
    return temporal_nearest_window<SEARCH>(X, Y, abc, n, j_prev, w.nn_back, w.nn_ahead);
  }
}

// run `body(tag)` with tag::value = the launch's search kind (one wave-uniform branch for the whole rollout)
template <typename Body>
__device__ __forceinline__ void with_search_kind(const Weights& w, int n, Body&& body, bool has_frames = false) {
  switch (search_kind(w, n, has_frames)) {
    case kSearchExhaustive: body(std::integral_constant<int, kSearchExhaustive>{}); break;
    case kSearchVerified: body(std::integral_constant<int, kSearchVerified>{}); break;
    case 8: body(std::integral_constant<int, 8>{}); break;
    case 4: body(std::integral_constant<int, 4>{}); break;
    default: body(std::integral_constant<int, kSearchGeneric>{}); break;
  }
}

// per-step dispatch, for the kernels that roll out one trajectory per wave or tile (finalize, sampled, tile)
__device__ __forceinline__ int step_temporal(StateT& s, const float* wp, const float* abc, int n, float v, float k,
                                             const Weights& w, int j_prev) {
  int j = 0;
  with_search_kind(w, n, [&](auto kind) {
    j = step_temporal_as<decltype(kind)::value>(s, wp, abc, n, v, k, w, j_prev);
  });
  return j;
}

template <typename F>
__device__ __forceinline__ F finish_temporal(const StateT_<F>& s, int n, const Weights& w) {
  const float tN = static_cast<float>(n) * w.dt;
  // J = 1/2 (q0 sum e_y^2 + q1 sum e_psi^2 + r0 sum dv^2 + r1 sum dkappa^2) with the halved weights, then the terminal terms
  F stage = splat<F>(w.hq0) * s.S0;
  stage = fma_(splat<F>(w.hq1), s.S1, stage);
  stage = fma_(splat<F>(w.hr0), s.S2, stage);
  stage = fma_(splat<F>(w.hr1), s.S3, stage);
  F a = (w.hqn0 * s.ey) * s.ey;
  a = fma_(w.hqn1 * s.ep, s.ep, a);
  a = fma_(splat<F>(w.hqn2 * tN), splat<F>(tN), a);
  const F J = stage + a;
  return fma_(splat<F>(w.wbound), s.V, J);
}

// ---- candidate sampling ------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based, so candidate c
// of problem p in round r is the same numbers on every rank and in every launch shape.
__device__ __host__ __forceinline__ void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// 32 random bits -> uniform in (0, 1): 24 bits, centred, never 0 or 1
__device__ __forceinline__ float uniform_open(uint32_t bits) {
  return static_cast<float>(bits >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
}

// ---- the sampler's normals, specified bit for bit (round 4) -------------------------------------------------------
// Box-Muller on two uniforms -> two standard normals with a FIXED operation sequence, so that oracle sample_candidates()
// reproduces every candidate of acmpc_sample_device / the fused rounds exactly, on any part (rounds 1-3 used v_log_f32 /
// v_sin_f32 / v_cos_f32: ~1e-6 apart from any restatement, and not portable):
//   ln u      u = m 2^e with m in [sqrt(1/2), sqrt(2)) read off the bit pattern, f = m - 1 (exact),
//             ln u = f - f^2/2 + f^3 P(f) + e ln 2 with the degree-8 P of Cephes' logf (Moshier) and ln 2 in two parts,
//             every multiply-add below ONE fmaf.  u is a uniform_open() value: normal, in (0, 1].
//   radius    sqrt(-2 ln u): IEEE multiply and correctly rounded square root (hipcc's default for float32).
//   angle     sin / cos of 2 pi u2 reduced in TURNS: k = rint(2 u2) off a magic-number sum, r = u2 - k/2 (exact),
//             x = 2 pi r in [-pi/2, pi/2], the rollout's own polynomials (kSinC, kCosC), sign (-1)^k.
constexpr float kLogC[9] = {7.0376836292e-2f, -1.1514610310e-1f, 1.1676998740e-1f, -1.2420140846e-1f, 1.4249322787e-1f,
                            -1.6668057665e-1f, 2.0000714765e-1f, -2.4999993993e-1f, 3.3333331174e-1f};
constexpr float kLn2Hi = 0.693359375f;         // 9 significant bits: e * kLn2Hi is exact
constexpr float kLn2Lo = -2.12194440e-4f;      // ln 2 - kLn2Hi

__device__ __forceinline__ float log_spec(float u) {
  const int bits = __float_as_int(u);
  const int e = (bits - 0x3f3504f3) >> 23;                 // arithmetic shift: floor(log2(u / sqrt(1/2)))
  const float m = __int_as_float(bits - (e << 23));        // u / 2^e in [sqrt(1/2), sqrt(2))
  const float f = m - 1.0f;
  const float ef = static_cast<float>(e);
  const float z = f * f;
  float p = fma_(kLogC[0], f, kLogC[1]);
#pragma unroll
  for (int q = 2; q < 9; ++q) p = fma_(p, f, kLogC[q]);
  float y = (f * z) * p;
  y = fma_(ef, kLn2Lo, y);
  y = fma_(-0.5f, z, y);
  return fma_(ef, kLn2Hi, f + y);
}

__device__ __forceinline__ void box_muller(float u1, float u2, float& z0, float& z1) {
  const float radius = __builtin_sqrtf(-2.0f * log_spec(u1));
  const float t = fma_(u2, 2.0f, 12582912.0f);             // k = rint(2 u2) in the low mantissa bits
  const float k = t - 12582912.0f;
  const float r = fma_(k, -0.5f, u2);                       // exact
  const float x = r * 6.28318530717959f;
  const float x2 = x * x;
  float ps = fma_(x2, kSinC[3], kSinC[2]);
  ps = fma_(x2, ps, kSinC[1]);
  ps = fma_(x2, ps, kSinC[0]);
  const float sn = fma_(x * x2, ps, x);
  float pc = fma_(x2, kCosC[4], kCosC[3]);
  pc = fma_(x2, pc, kCosC[2]);
  pc = fma_(x2, pc, kCosC[1]);
  pc = fma_(x2, pc, kCosC[0]);
  const float cs = fma_(x2, pc, 1.0f);
  const int sign = __float_as_int(t) << 31;
  z0 = radius * __int_as_float(__float_as_int(cs) ^ sign);
  z1 = radius * __int_as_float(__float_as_int(sn) ^ sign);
}

constexpr int kKnots = 8;  // raised-cosine knots along the horizon (== kSampleKnots)
constexpr int kKnotsMax = kKnots;

// Everything that defines candidate `gidx` of problem `p` in round `round` besides its centre.
struct SampleSpec {
  const float* segments;  // [n][2]: left knot (as float), weight of the left knot
  int knot_begin[kKnotsMax + 1];  // steps [knot_begin[k], knot_begin[k+1]) have left knot k (kernel argument: SGPRs)
  uint32_t seed_lo, seed_hi, round;
  const uint32_t* seed_ptr;  // when non-null the key is read from device memory (two words) instead of seed_lo/hi:
                             // lets a captured hipGraph be replayed with a new seed without touching its nodes
  float sigma_v, sigma_k;
  float ulo0, ulo1, uhi0, uhi1;
};

// the 8 x 2 standard normals of one candidate; draws Q0 .. Q1 - 1 of the kKnots / 2 (a caller with something to wait
// for in between takes them in two halves)
// one Philox block of a candidate's normals: the four of knots 2q and 2q + 1 - out = {z[2q][0], z[2q][1], z[2q+1][0],
// z[2q+1][1]} as draw_normals() below fills them (q may be a run-time value: lanes that share a candidate share the draws)
__device__ __forceinline__ void draw_normal_block(const SampleSpec& sp, uint32_t gidx, uint32_t p, uint32_t q, float (&out)[4]) {
  const uint32_t key[2] = {sp.seed_ptr != nullptr ? sp.seed_ptr[0] : sp.seed_lo,
                           sp.seed_ptr != nullptr ? sp.seed_ptr[1] : sp.seed_hi};
  const uint32_t ctr[4] = {gidx, p, sp.round, q};
  uint32_t r[4];
  philox4x32_10(ctr, key, r);
  box_muller(uniform_open(r[0]), uniform_open(r[1]), out[0], out[1]);
  box_muller(uniform_open(r[2]), uniform_open(r[3]), out[2], out[3]);
}

template <int Q0 = 0, int Q1 = kKnots / 2>
__device__ __forceinline__ void draw_normals(const SampleSpec& sp, uint32_t gidx, uint32_t p, float (&z)[kKnots][2]) {
  const uint32_t key[2] = {sp.seed_ptr != nullptr ? sp.seed_ptr[0] : sp.seed_lo,
                           sp.seed_ptr != nullptr ? sp.seed_ptr[1] : sp.seed_hi};
#pragma unroll
  for (int q = Q0; q < Q1; ++q) {
    const uint32_t ctr[4] = {gidx, p, sp.round, static_cast<uint32_t>(q)};
    uint32_t r[4];
    philox4x32_10(ctr, key, r);
    box_muller(uniform_open(r[0]), uniform_open(r[1]), z[2 * q][0], z[2 * q][1]);
    box_muller(uniform_open(r[2]), uniform_open(r[3]), z[2 * q + 1][0], z[2 * q + 1][1]);
  }
}

__device__ __forceinline__ float candidate_amplitude(uint32_t gidx) {
  return (gidx == 0u) ? 0.0f : static_cast<float>((gidx & 7u) + 1u) * 0.125f;
}

// control (v, kappa) of one candidate at one step from the normals of the two knots that bracket it
__device__ __forceinline__ void blend_control(const SampleSpec& sp, float amp, float w0, float cv, float ck, float z0v,
                                              float z0k, float z1v, float z1k, float& v, float& k) {
  const float w1 = 1.0f - w0;
  v = cv + (sp.sigma_v * amp) * (w0 * z0v + w1 * z1v);
  k = ck + (sp.sigma_k * amp) * (w0 * z0k + w1 * z1k);
  v = fminf(fmaxf(v, sp.ulo0), sp.uhi0);
  k = fminf(fmaxf(k, sp.ulo1), sp.uhi1);
}

// ---- (cost, index) keys ---------------------------------------------------------------------------------
// key = (ordered_int32(cost) << 32) | uint32(index): signed 64-bit order == (cost, index) lexicographic order,
// so min() is np.argmin's "first minimum".  Non-finite costs rank as +inf.
__device__ __host__ __forceinline__ int64_t pack_key(float cost, uint32_t index) {
  union {
    float f;
    int32_t i;
    uint32_t u;
  } b;
  b.f = cost;
  if ((b.u & 0x7f800000u) == 0x7f800000u) b.u = 0x7f800000u;  // inf / nan -> +inf
  const int32_t hi = (b.i >= 0) ? b.i : (b.i ^ 0x7fffffff);
  return (static_cast<int64_t>(hi) << 32) | static_cast<int64_t>(index);
}

constexpr int64_t kKeyMax = INT64_MAX;

// Wave reductions on the DPP path (row shifts inside the rows of 16 lanes, then row_bcast:15 / :31 across them; the
// result is lane 63's): six vector instructions per 32-bit reduction where the shuffle forms (ds_bpermute) make six round
// trips through the LDS crossbar - half a microsecond for a key + a count on a lone wave.  Integers: exact either way.
// In place: v = op(v shifted within its row, v) - lanes the shift has no source for keep their value (no bound_ctrl), which
// is what a scan wants.  Inline assembly because the compiler does not fold update_dpp into the consumer here (it emits
// v_mov + s_nop + v_mov_dpp + op per stage); the two wait states a DPP read needs after a vector write are spelt out, and
// one more pair after the last stage for whatever reads the result next (the hazard recogniser does not look in here).
#define ACMPC_DPP_SCAN(op)                                                         \
  "s_nop 1\n" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"             \
  "s_nop 1\n" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"             \
  "s_nop 1\n" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"             \
  "s_nop 1\n" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"             \
  "s_nop 1\n" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"          \
  "s_nop 1\n" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"          \
  "s_nop 1"

__device__ __forceinline__ int wave_sum_int(int v) {
  asm volatile(ACMPC_DPP_SCAN("v_add_u32_dpp") : "+v"(v));   // lane 63 holds the wave's sum
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int wave_min_int(int v) {
  asm volatile(ACMPC_DPP_SCAN("v_min_i32_dpp") : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// min over the wave's keys = lexicographic (signed cost word, unsigned index word): the minimum of the cost words, then
// the minimum of the index words (sign bit flipped for the signed instruction) among the lanes that hold that cost
__device__ __forceinline__ int64_t wave_min_key(int64_t v) {
  const int hi = static_cast<int>(v >> 32);
  const int best_hi = wave_min_int(hi);
  const int lo = (hi == best_hi) ? static_cast<int>(static_cast<uint32_t>(v & 0xffffffffLL) ^ 0x80000000u) : INT32_MAX;
  const uint32_t best_lo = static_cast<uint32_t>(wave_min_int(lo)) ^ 0x80000000u;
  return (static_cast<int64_t>(best_hi) << 32) | best_lo;
}

// the same when the index rises with the lane (index = base + lane): the minimum of the cost words, then the FIRST lane
// that holds it.  `lane_out` = that lane.
__device__ __forceinline__ int64_t wave_min_key_by_lane(int64_t own_key, int& lane_out) {
  const int hi = static_cast<int>(own_key >> 32);
  const int best = wave_min_int(hi);
  const unsigned long long holders = __ballot(hi == best);
  lane_out = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(holders)) - 1);
  const int lo = __builtin_amdgcn_readlane(static_cast<int>(own_key & 0xffffffffLL), lane_out);
  return (static_cast<int64_t>(best) << 32) | static_cast<uint32_t>(lo);
}

}  // namespace acmpc
