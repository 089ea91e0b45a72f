/*
 * acmpc_oracle.c - scalar C restatement of the sampling composition (TEST INFRASTRUCTURE ONLY).
 *
 * Same float32 "spec order" as oracle/acmpc_oracle.py (rollout_spatial / rollout_temporal): one fixed
 * association, no implicit FMA contraction (-ffp-contract=off), no libm transcendentals in the rollouts; mode T's
 * specification names its fused multiply-adds explicitly and they are fmaf() here (IEEE, one rounding).  It exists to
 * (a) cross-check the NumPy oracle bit-for-bit at sizes NumPy finishes slowly, and (b) serve as the CPU baseline
 * ("port") that bench.py times beside the GPU.  Nothing under ac-mpc_amd/ links or loads it.
 *
 * Reference arithmetic restated (paths relative to /root/reference/src/acmpc/):
 *   step_spatial  : x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i     control/dynamics.py:65-103,
 *                   cost 1/2 (x'Qx + du'R du), bounds                  control/solvers/control.py:26-79,121-158
 *   step_temporal : kinematic Euler step                               localisation/localiser.py:66-95
 *                   nearest waypoint (first minimum)                   localisation/localiser.py:282-289
 *                   Frenet errors                                      control/dynamics.py:23-40
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>

#define CS 12 /* floats per mode-S row: ds, a21, a31, b31, f3, v_ref, k_ref, ey_lo, ey_hi, 0, 0, 0 */
#define CT 8  /* floats per mode-T row: x, y, cos psi, sin psi, psi, k_ref, v_ref, w/2 - margin    */

typedef struct {
  float q[3], r[2], qn[3], ulo[2], uhi[2], tmin, wbound, dt;
  int nn_back, nn_ahead; /* mode T search window round the previous nearest index; nn_ahead < 0 = exhaustive */
} oracle_weights;

static inline float quad(float w, float a) { return (w * a) * a; }

static inline float hinge2(float lo_minus_x, float x_minus_hi) {
  const float v = fmaxf(fmaxf(lo_minus_x, x_minus_hi), 0.0f);
  return v * v;
}

static inline void fetch(const float* U, int layout, int64_t N, int n, int64_t c, int i, float* v, float* k) {
  if (layout == 0) { /* U[N][n][2] */
    const float* p = U + (c * n + i) * 2;
    *v = p[0];
    *k = p[1];
  } else { /* U[n][2][N] */
    *v = U[((int64_t)i * 2) * N + c];
    *k = U[((int64_t)i * 2 + 1) * N + c];
  }
}

/* costs[N], viol[N]; states (optional) [N][n+1][3].  Returns nothing; threads over candidates. */
void acmpc_oracle_rollout_spatial(const float* x0, const float* coef, const float* U, int layout, int64_t N, int n,
                                  const oracle_weights* w, float* costs, float* viol, float* states) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < N; ++c) {
    float ey = x0[0], ep = x0[1], t = x0[2], J = 0.0f, V = 0.0f;
    for (int i = 0; i < n; ++i) {
      const float* k_ = coef + (int64_t)i * CS;
      float v, k;
      fetch(U, layout, N, n, c, i, &v, &k);
      if (states) {
        float* s = states + (c * (n + 1) + i) * 3;
        s[0] = ey, s[1] = ep, s[2] = t;
      }
      const float dv = v - k_[5];
      const float dk = k - k_[6];
      float a = quad(w->q[0], ey);
      a = a + quad(w->q[1], ep);
      a = a + quad(w->q[2], t);
      float r = quad(w->r[0], dv);
      r = r + quad(w->r[1], dk);
      J = J + 0.5f * (a + r);
      V = V + hinge2(w->ulo[0] - v, v - w->uhi[0]);
      V = V + hinge2(w->ulo[1] - k, k - w->uhi[1]);
      const float ey_n = ey + k_[0] * ep;
      const float ep_n = (ep + k_[1] * ey) + k_[0] * dk;
      const float t_n = ((t + k_[2] * ey) + k_[3] * dv) + k_[4];
      ey = ey_n, ep = ep_n, t = t_n;
      V = V + hinge2(k_[7] - ey, ey - k_[8]);
      const float tv = fmaxf(w->tmin - t, 0.0f);
      V = V + tv * tv;
    }
    if (states) {
      float* s = states + (c * (n + 1) + n) * 3;
      s[0] = ey, s[1] = ep, s[2] = t;
    }
    float a = quad(w->qn[0], ey);
    a = a + quad(w->qn[1], ep);
    a = a + quad(w->qn[2], t);
    J = J + 0.5f * a;
    costs[c] = J + w->wbound * V;
    viol[c] = V;
  }
}

/* Same arithmetic as acmpc_oracle_rollout_spatial for the step-major layout U[n][2][N], organised so that the
 * compiler can vectorise across candidates (blocks of VB candidates advance through the steps together).  This is
 * the form bench.py times as the CPU baseline; results are bit-identical to the scalar form (IEEE single, no FMA). */
/* a > b ? a : b equals fmaxf(a, b) whenever b is not NaN - true at both call sites (b is x - hi with the same x
 * as a, or the constant 0) - and unlike fmaxf it vectorises without -ffinite-math-only */
static inline float max2(float a, float b) { return a > b ? a : b; }
static inline float hinge2v(float lo_minus_x, float x_minus_hi) {
  const float v = max2(max2(lo_minus_x, x_minus_hi), 0.0f);
  return v * v;
}

#define VB 64
/* one block of VB candidates through all steps; a plain function with restrict-qualified pointers and the weights
 * in locals, so that the compiler can prove the lanes independent (inside the OpenMP-outlined region it cannot) */
static void spatial_block(const float* restrict x0, const float* restrict coef, const float* restrict U, int64_t N,
                          int n, const oracle_weights* restrict wp, int64_t c0, int m, float* restrict costs,
                          float* restrict viol) {
  const oracle_weights w = *wp;
  float ey[VB], ep[VB], t[VB], J[VB], V[VB];
  for (int j = 0; j < VB; ++j) ey[j] = x0[0], ep[j] = x0[1], t[j] = x0[2], J[j] = 0.0f, V[j] = 0.0f;
  for (int i = 0; i < n; ++i) {
    const float* restrict k_ = coef + (int64_t)i * CS;
    const float c_ds = k_[0], c_a21 = k_[1], c_a31 = k_[2], c_b31 = k_[3], c_f3 = k_[4], c_vref = k_[5],
                c_kref = k_[6], c_lo = k_[7], c_hi = k_[8];
    const float* restrict vrow = U + ((int64_t)i * 2) * N + c0;
    const float* restrict krow = vrow + N;
#pragma omp simd
    for (int j = 0; j < m; ++j) {
      const float v = vrow[j], k = krow[j];
      const float dv = v - c_vref;
      const float dk = k - c_kref;
      float a = quad(w.q[0], ey[j]);
      a = a + quad(w.q[1], ep[j]);
      a = a + quad(w.q[2], t[j]);
      float r = quad(w.r[0], dv);
      r = r + quad(w.r[1], dk);
      J[j] = J[j] + 0.5f * (a + r);
      float Vj = V[j] + hinge2v(w.ulo[0] - v, v - w.uhi[0]);
      Vj = Vj + hinge2v(w.ulo[1] - k, k - w.uhi[1]);
      const float ey_n = ey[j] + c_ds * ep[j];
      const float ep_n = (ep[j] + c_a21 * ey[j]) + c_ds * dk;
      const float t_n = ((t[j] + c_a31 * ey[j]) + c_b31 * dv) + c_f3;
      ey[j] = ey_n, ep[j] = ep_n, t[j] = t_n;
      Vj = Vj + hinge2v(c_lo - ey_n, ey_n - c_hi);
      const float tv = max2(w.tmin - t_n, 0.0f);
      V[j] = Vj + tv * tv;
    }
  }
  for (int j = 0; j < m; ++j) {
    float a = quad(w.qn[0], ey[j]);
    a = a + quad(w.qn[1], ep[j]);
    a = a + quad(w.qn[2], t[j]);
    const float Jf = J[j] + 0.5f * a;
    costs[c0 + j] = Jf + w.wbound * V[j];
    viol[c0 + j] = V[j];
  }
}

void acmpc_oracle_rollout_spatial_blocked(const float* x0, const float* coef, const float* U, int64_t N, int n,
                                          const oracle_weights* w, float* costs, float* viol) {
  const int64_t blocks = (N + VB - 1) / VB;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < blocks; ++b) {
    const int64_t c0 = b * VB;
    const int m = (int)((N - c0) < VB ? (N - c0) : VB);
    spatial_block(x0, coef, U, N, n, w, c0, m, costs, viol);
  }
}

/* P problems at once (x0 [P][3], coef [P][n][CS], U [P][n][2][N], costs/viol [P][N]): one parallel region over all
 * (problem, block) pairs, so that the threads stay busy when a single problem is only a few dozen blocks */
void acmpc_oracle_rollout_spatial_batch(const float* x0, const float* coef, const float* U, int P, int64_t N, int n,
                                        const oracle_weights* w, float* costs, float* viol) {
  const int64_t blocks = (N + VB - 1) / VB;
#pragma omp parallel for schedule(static)
  for (int64_t job = 0; job < (int64_t)P * blocks; ++job) {
    const int64_t p = job / blocks, b = job % blocks;
    const int64_t c0 = b * VB;
    const int m = (int)((N - c0) < VB ? (N - c0) : VB);
    spatial_block(x0 + p * 3, coef + p * (int64_t)n * CS, U + p * (int64_t)n * 2 * N, N, n, w, c0, m, costs + p * N,
                  viol + p * N);
  }
}

/* rint and quadrant without a float -> int conversion (out-of-range conversions differ between platforms):
 * t = y + 1.5 * 2^23 carries rint(y) in its low mantissa bits for |y| < 2^22 */
static inline int32_t float_bits(float f) {
  union {
    float f;
    int32_t i;
  } b;
  b.f = f;
  return b.i;
}

/* phi = k pi + r, k = rint(phi / pi) read from the mantissa of t, sin phi = (-1)^k sin r, cos phi = (-1)^k cos r */
static inline void sincos_spec(float phi, float* sn, float* cs) {
  const float t = fmaf(phi, 0.3183098861837907f, 12582912.0f);
  const float k = t - 12582912.0f;
  float r = fmaf(-k, 3.140625f, phi);
  r = fmaf(-k, 9.67653589793e-4f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, 2.59990065387683e-06f, -0.00019806546333711594f);
  ps = fmaf(r2, ps, 0.008333016186952591f);
  ps = fmaf(r2, ps, -0.16666656732559204f);
  const float s = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, -2.607563374112942e-07f, 2.4761806344031356e-05f);
  pc = fmaf(r2, pc, -0.0013888402609154582f);
  pc = fmaf(r2, pc, 0.04166664183139801f);
  pc = fmaf(r2, pc, -0.5f);
  const float c = fmaf(r2, pc, 1.0f);
  const int odd = float_bits(t) & 1;
  *sn = odd ? -s : s;
  *cs = odd ? -c : c;
}

/* a - 2 pi rint(a / 2 pi), the integer read off the magic-number sum (csrc/acmpc_device.h: wrap_spec) */
static inline float wrap_spec(float a) {
  const float q = fmaf(a, 0.159154943091895f, 12582912.0f) - 12582912.0f;
  return fmaf(-q, 6.28318530717959f, a);
}

void acmpc_oracle_rollout_temporal(const float* pose0, const float* wp, const float* U, int layout, int64_t N, int n,
                                   const oracle_weights* w, float* costs, float* viol, float* states) {
  /* the halved weights of the term-by-term accumulation J += (w/2 a) a */
  const float hq0 = 0.5f * w->q[0], hq1 = 0.5f * w->q[1], hr0 = 0.5f * w->r[0], hr1 = 0.5f * w->r[1];
  const float hqn0 = 0.5f * w->qn[0], hqn1 = 0.5f * w->qn[1], hqn2 = 0.5f * w->qn[2];
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < N; ++c) {
    /* the path's own frame (csrc/acmpc_device.h: start_temporal): positions relative to waypoint 0, float32 differences */
    const float ox = wp[0], oy = wp[1];
    float X = pose0[0] - ox, Y = pose0[1] - oy, phi = pose0[2], ey = 0.0f, ep = 0.0f, V = 0.0f;
    float S0 = 0.0f, S1 = 0.0f, S2 = 0.0f, S3 = 0.0f; /* sums of e_y^2, e_psi^2, dv^2, dkappa^2: weights applied at the end */
    int j_prev = 0;
    if (states) {
      float* s = states + c * (n + 1) * 3;
      s[0] = X + ox, s[1] = Y + oy, s[2] = phi;
    }
    for (int i = 0; i < n; ++i) {
      float v, k, sn, cs;
      fetch(U, layout, N, n, c, i, &v, &k);
      sincos_spec(phi, &sn, &cs);
      const float Xn = fmaf(v * cs, w->dt, X);
      const float Yn = fmaf(v * sn, w->dt, Y);
      const float phin = fmaf(v * k, w->dt, phi);
      X = Xn, Y = Yn, phi = phin;
      float best = INFINITY;
      int j = 0;
      int lo = 0, hi = n - 1;
      if (w->nn_ahead >= 0) { /* W consecutive waypoints from clamp(j_prev - back, 0, n - W) */
        const int width = w->nn_back + w->nn_ahead + 1;
        lo = j_prev - w->nn_back;
        if (lo > n - width) lo = n - width;
        if (lo < 0) lo = 0;
        hi = lo + width > n ? n - 1 : lo + width - 1;
        j = lo;
      }
      for (int m = lo; m <= hi; ++m) {
        /* search key (csrc/acmpc_device.h: search_key): |p - w_m|^2 less |p|^2, two fused multiply-adds */
        const float wx = wp[m * CT + 0] - ox, wy = wp[m * CT + 1] - oy;
        const float d = fmaf(Y, -2.0f * wy, fmaf(X, -2.0f * wx, fmaf(wy, wy, wx * wx)));
        if (d < best) {
          best = d;
          j = m;
        }
      }
      j_prev = j;
      const float* g = wp + j * CT;
      /* e_y = c (Y - y) - s (X - x) as the kernels' derived rows evaluate it: fma(c, Y, fma(-s, X, s x - c y)) */
      ey = fmaf(g[2], Y, fmaf(-g[3], X, fmaf(g[3], g[0] - ox, -(g[2] * (g[1] - oy)))));
      ep = wrap_spec(phi - g[4]);
      const float dv = v - g[6];
      const float dk = k - g[5];
      S0 = fmaf(ey, ey, S0);
      S1 = fmaf(ep, ep, S1);
      S2 = fmaf(dv, dv, S2);
      S3 = fmaf(dk, dk, S3);
      /* excess over the input box: x - med3(x, lo, hi) */
      const float hv = v - fminf(fmaxf(v, w->ulo[0]), w->uhi[0]);
      V = fmaf(hv, hv, V);
      const float hk = k - fminf(fmaxf(k, w->ulo[1]), w->uhi[1]);
      V = fmaf(hk, hk, V);
      const float hc = fmaxf(fabsf(ey) - g[7], 0.0f);
      V = fmaf(hc, hc, V);
      if (states) {
        float* s = states + (c * (n + 1) + i + 1) * 3;
        s[0] = X + ox, s[1] = Y + oy, s[2] = phi;
      }
    }
    const float tN = (float)n * w->dt;
    float a = (hqn0 * ey) * ey;
    a = fmaf(hqn1 * ep, ep, a);
    a = fmaf(hqn2 * tN, tN, a);
    float J = hq0 * S0;
    J = fmaf(hq1, S1, J);
    J = fmaf(hr0, S2, J);
    J = fmaf(hr1, S3, J);
    J = J + a;
    costs[c] = fmaf(w->wbound, V, J);
    viol[c] = V;
  }
}

/* first minimum; non-finite costs rank as +inf */
int64_t acmpc_oracle_argmin(const float* costs, int64_t N) {
  int64_t best = 0;
  float bc = INFINITY;
  for (int64_t c = 0; c < N; ++c) {
    const float v = isfinite(costs[c]) ? costs[c] : INFINITY;
    if (v < bc) {
      bc = v;
      best = c;
    }
  }
  return best;
}
