// Device-side arithmetic of the rollout-and-cost path, gfx950 only.
//
// Every function here is the float32 "spec order" of DESIGN.md: one fixed association, no FMA contraction
// (the translation unit is built with -ffp-contract=off and the pragma below), no library transcendentals.
// That is what makes costs bit-identical to oracle/acmpc_oracle.{py,c}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace acmpc {

constexpr int kWave = 64;
constexpr int kCoefS = 12;  // ACMPC_COEF_STRIDE_SPATIAL
constexpr int kCoefT = 8;   // ACMPC_COEF_STRIDE_TEMPORAL

// Scalars of the cost and the bounds; passed by value in the kernel argument segment, so they live in SGPRs.
struct Weights {
  float q0, q1, q2;     // Q   (control.py:126)
  float r0, r1;         // R   (control.py:127)
  float qn0, qn1, qn2;  // QN  (control.py:128)
  float ulo0, ulo1, uhi0, uhi1;  // input box incl. the 0.1 m/s slack (control.py:130-139)
  float tmin;           // 0.01 (control.py:134)
  float wbound;
  float dt;
  int nn_back, nn_ahead;  // mode T search window round the previous nearest index; nn_ahead < 0 = exhaustive
};

__device__ __forceinline__ float quad(float w, float a) { return (w * a) * a; }

__device__ __forceinline__ float hinge2(float lo_minus_x, float x_minus_hi) {
  // at most one side of a (non-degenerate) interval can be violated: one v_max3_f32
  const float v = fmaxf(fmaxf(lo_minus_x, x_minus_hi), 0.0f);
  return v * v;
}

// ---- mode S --------------------------------------------------------------------------------------------
// One step of x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i (dynamics.py:65-103, control.py:26-45) with the
// stage cost 1/2 (x'Qx + du'R du) (control.py:72-79,151-158) and the squared violation of the input box, the
// corridor of x_{i+1} (control.py:57-60) and t >= t_min (control.py:134).
struct StateS {
  float ey, ep, t, J, V;
};

__device__ __forceinline__ void step_spatial(StateS& s, const float* __restrict__ c, float v, float k,
                                             const Weights& w) {
  const float dv = v - c[5];
  const float dk = k - c[6];
  float a = quad(w.q0, s.ey);
  a = a + quad(w.q1, s.ep);
  a = a + quad(w.q2, s.t);
  float r = quad(w.r0, dv);
  r = r + quad(w.r1, dk);
  s.J = s.J + 0.5f * (a + r);
  s.V = s.V + hinge2(w.ulo0 - v, v - w.uhi0);
  s.V = s.V + hinge2(w.ulo1 - k, k - w.uhi1);
  const float ds = c[0];
  const float ey_n = s.ey + ds * s.ep;
  const float ep_n = (s.ep + c[1] * s.ey) + ds * dk;
  const float t_n = ((s.t + c[2] * s.ey) + c[3] * dv) + c[4];
  s.ey = ey_n;
  s.ep = ep_n;
  s.t = t_n;
  s.V = s.V + hinge2(c[7] - s.ey, s.ey - c[8]);
  const float tv = fmaxf(w.tmin - s.t, 0.0f);
  s.V = s.V + tv * tv;
}

__device__ __forceinline__ float finish_spatial(const StateS& s, const Weights& w) {
  float a = quad(w.qn0, s.ey);
  a = a + quad(w.qn1, s.ep);
  a = a + quad(w.qn2, s.t);
  const float J = s.J + 0.5f * a;
  return J + w.wbound * s.V;
}

// ---- mode T --------------------------------------------------------------------------------------------
// Cody-Waite reduction by pi/2 and the Cephes single-precision minimax polynomials: the same instruction
// sequence as oracle sincos_spec().
__device__ __forceinline__ void sincos_spec(float phi, float& sn, float& cs) {
  const float k = rintf(phi * 0.6366197723675814f);
  const float r = (phi - k * 1.5703125f) - k * 4.838267948966e-4f;
  const float r2 = r * r;
  float ps = 8.3321608736e-3f + r2 * -1.9515295891e-4f;
  ps = -1.6666654611e-1f + r2 * ps;
  const float s = r + (r * r2) * ps;
  float pc = -1.388731625493765e-3f + r2 * 2.443315711809948e-5f;
  pc = 4.166664568298827e-2f + r2 * pc;
  const float c = (1.0f - 0.5f * r2) + (r2 * r2) * pc;
  const int q = static_cast<int>(k) & 3;
  sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
  cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

__device__ __forceinline__ float wrap_spec(float a) {
  const float b = a + 3.14159265358979f;
  const float q = floorf(b * 0.159154943091895f);
  return (b - q * 6.28318530717959f) - 3.14159265358979f;
}

struct StateT {
  float X, Y, phi, ey, ep, J, V;
};

// explicit Euler on the rear-axle kinematic bicycle (localiser.py:66-95); phi_dot = v * kappa
__device__ __forceinline__ void temporal_advance(StateT& s, float v, float k, const Weights& w) {
  float sn, cs;
  sincos_spec(s.phi, sn, cs);
  const float Xn = s.X + (v * cs) * w.dt;
  const float Yn = s.Y + (v * sn) * w.dt;
  const float phin = s.phi + (v * k) * w.dt;
  s.X = Xn;
  s.Y = Yn;
  s.phi = phin;
}

__device__ __forceinline__ float dist2(float X, float Y, float wx, float wy) {
  const float dx = X - wx;
  const float dy = Y - wy;
  return dx * dx + dy * dy;
}

// nearest waypoint, first minimum of the squared distance (localiser.py:282-289); one lane scans the table
__device__ __forceinline__ int temporal_nearest(const StateT& s, const float* wp, int n) {
  float best = __builtin_inff();
  int j = 0;
  for (int i = 0; i < n; ++i) {
    const float d = dist2(s.X, s.Y, wp[i * kCoefT + 0], wp[i * kCoefT + 1]);
    const bool better = d < best;
    best = better ? d : best;
    j = better ? i : j;
  }
  return j;
}

// The same search restricted to [j_prev - back, j_prev + ahead] (clipped to the table): progress along the path is
// monotone and at most about one waypoint per step, so a short window finds the global minimum on every realistic
// input at a fraction of the ALU work (tests check equality with the exhaustive scan).  Fixed trip count, lanes
// whose window is clipped mask the tail.
__device__ __forceinline__ int temporal_nearest_window(const StateT& s, const float* wp, int n, int j_prev,
                                                       int back, int ahead) {
  const int lo = max(j_prev - back, 0);
  const int hi = min(j_prev + ahead, n - 1);
  float best = __builtin_inff();
  int j = lo;
  const int trips = back + ahead + 1;
  for (int m = 0; m < trips; ++m) {
    const int i = min(lo + m, hi);  // the clipped tail re-reads `hi`: equal distance, never "better"
    const float d = dist2(s.X, s.Y, wp[i * kCoefT + 0], wp[i * kCoefT + 1]);
    const bool better = d < best;
    best = better ? d : best;
    j = better ? i : j;
  }
  return j;
}

// Frenet errors w.r.t. waypoint row g (dynamics.py:23-40), stage cost and bound violations
__device__ __forceinline__ void temporal_cost(StateT& s, const float* g, float v, float k, const Weights& w) {
  s.ey = g[2] * (s.Y - g[1]) - g[3] * (s.X - g[0]);
  s.ep = wrap_spec(s.phi - g[4]);
  const float dv = v - g[6];
  const float dk = k - g[5];
  float a = quad(w.q0, s.ey);
  a = a + quad(w.q1, s.ep);
  float r = quad(w.r0, dv);
  r = r + quad(w.r1, dk);
  s.J = s.J + 0.5f * (a + r);
  s.V = s.V + hinge2(w.ulo0 - v, v - w.uhi0);
  s.V = s.V + hinge2(w.ulo1 - k, k - w.uhi1);
  s.V = s.V + hinge2((-g[7]) - s.ey, s.ey - g[7]);
}

// `wp` is the waypoint table (kCoefT floats per waypoint); in the rollout kernel it lives in LDS.
__device__ __forceinline__ int step_temporal(StateT& s, const float* wp, int n, float v, float k,
                                             const Weights& w, int j_prev) {
  temporal_advance(s, v, k, w);
  const int j = (w.nn_ahead < 0) ? temporal_nearest(s, wp, n)
                                 : temporal_nearest_window(s, wp, n, j_prev, w.nn_back, w.nn_ahead);
  temporal_cost(s, wp + j * kCoefT, v, k, w);
  return j;
}

__device__ __forceinline__ float finish_temporal(const StateT& s, int n, const Weights& w) {
  const float tN = static_cast<float>(n) * w.dt;
  float a = quad(w.qn0, s.ey);
  a = a + quad(w.qn1, s.ep);
  a = a + quad(w.qn2, tN);
  const float J = s.J + 0.5f * a;
  return J + w.wbound * s.V;
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

// Box-Muller on two uniforms -> two standard normals
__device__ __forceinline__ void box_muller(float u1, float u2, float& z0, float& z1) {
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.28318530717958647692f * u2, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

constexpr int kKnots = 8;  // raised-cosine knots along the horizon (== kSampleKnots)

// Everything that defines candidate `gidx` of problem `p` in round `round` besides its centre.
struct SampleSpec {
  const float* segments;  // [n][2]: left knot (as float), weight of the left knot
  uint32_t seed_lo, seed_hi, round;
  float sigma_v, sigma_k;
  float ulo0, ulo1, uhi0, uhi1;
};

// the 8 x 2 standard normals of one candidate
__device__ __forceinline__ void draw_normals(const SampleSpec& sp, uint32_t gidx, uint32_t p, float (&z)[kKnots][2]) {
  const uint32_t key[2] = {sp.seed_lo, sp.seed_hi};
#pragma unroll
  for (int q = 0; q < kKnots / 2; ++q) {
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

__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int mask) {
  int lo = static_cast<int>(v & 0xffffffffLL);
  int hi = static_cast<int>(v >> 32);
  lo = __shfl_xor(lo, mask, kWave);
  hi = __shfl_xor(hi, mask, kWave);
  return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}

__device__ __forceinline__ int64_t wave_min_key(int64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int64_t o = shfl_xor_i64(v, m);
    v = (o < v) ? o : v;
  }
  return v;
}

__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

}  // namespace acmpc
