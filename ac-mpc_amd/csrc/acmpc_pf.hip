// Particle scoring of the localiser on the GPU: the deterministic, data-parallel part of
// LocalisationProcess._update_particles (src/acmpc/localisation/localiser.py:255-410) plus the kinematic particle
// step (localiser.py:66-95) and the weighted-mean estimate with its convergence test (localiser.py:561-579).
// Resampling stays with the caller (it is sequential and random).
//
// One workgroup per particle.  Precision follows the reference step by step: float32 particle states and
// observations, placement of the observation in float32, a float64 map, float64 distances/error/score.  Nearest-
// neighbour indices must be the ones a KD-tree query returns, so the three exhaustive searches compare float64
// squared distances (lowest index on ties).  At the reference's sizes (500 particles, 1e4 map points, ~250
// observation points) the kernel is latency-bound; the map stays resident in device memory.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/acmpc.h"

namespace {

constexpr int kBlock = 256;
constexpr double kPi = 3.14159265358979323846;

struct Track {
  const double* xy;  // [m][2]
  int m;
};

struct ScoreArgs {
  const float* states;     // [P][3]
  const float* obs;        // [K_left + K_right][2], vehicle frame (x right, y forward)
  int k_left, k_right;
  Track centre, left, right;
  double mean, sigma, scale;
  double thr_rotation, thr_offset, thr_error;
  int32_t* track_indices;  // [P][3]
  double* minimum_offset;  // [P]
  double* heading_offset;  // [P]
  double* error;           // [P]
  double* score;           // [P]
  uint8_t* valid;          // [P]
};

// (distance^2, index) minimum over the workgroup, lowest index on ties; result broadcast through LDS
__device__ void block_argmin(double& d2, int& idx, double* s_d, int* s_i) {
  const int tid = threadIdx.x;
  s_d[tid] = d2;
  s_i[tid] = idx;
  __syncthreads();
  for (int half = kBlock / 2; half > 0; half >>= 1) {
    if (tid < half) {
      const double od = s_d[tid + half];
      const int oi = s_i[tid + half];
      if (od < s_d[tid] || (od == s_d[tid] && oi < s_i[tid])) {
        s_d[tid] = od;
        s_i[tid] = oi;
      }
    }
    __syncthreads();
  }
  d2 = s_d[0];
  idx = s_i[0];
  __syncthreads();
}

__device__ void nearest_on_track(const Track t, double px, double py, double& d2, int& idx, double* s_d, int* s_i) {
  double best = INFINITY;
  int best_i = 0x7fffffff;
  for (int m = threadIdx.x; m < t.m; m += kBlock) {
    const double dx = px - t.xy[2 * m], dy = py - t.xy[2 * m + 1];
    const double d = dx * dx + dy * dy;
    if (d < best) {  // ascending m per thread: the first minimum stays
      best = d;
      best_i = m;
    }
  }
  block_argmin(best, best_i, s_d, s_i);
  d2 = best;
  idx = best_i;
}

__global__ void __launch_bounds__(kBlock) pf_score_kernel(const ScoreArgs a) {
  __shared__ double s_d[kBlock];
  __shared__ int s_i[kBlock];
  const int p = blockIdx.x;
  const int tid = threadIdx.x;
  const double px = a.states[3 * p], py = a.states[3 * p + 1], phi = a.states[3 * p + 2];

  // three nearest-neighbour queries (localiser.py:282-289)
  double d_centre, d_tmp;
  int i_centre, i_left, i_right;
  nearest_on_track(a.centre, px, py, d_centre, i_centre, s_d, s_i);
  nearest_on_track(a.left, px, py, d_tmp, i_left, s_d, s_i);
  nearest_on_track(a.right, px, py, d_tmp, i_right, s_d, s_i);

  // observation placed in this particle's frame vs the map limits ahead of the nearest points (:330-410)
  // The reference places the observation in float32 (float32 states and observation, :330-353) and only then
  // subtracts the float64 map: do the same, so that the placed points round the way its do.
  const float phi32 = a.states[3 * p + 2], px32 = a.states[3 * p], py32 = a.states[3 * p + 1];
  const float angle = -phi32 + 1.57079632679489661923f;
  const float ca = cosf(angle), sa = sinf(angle);
  const int K = a.k_left + a.k_right;
  double sum = 0.0;
  for (int k = tid; k < K; k += kBlock) {
    const float ox = a.obs[2 * k], oy = a.obs[2 * k + 1];
    const double wx = (ca * ox + sa * oy) + px32;   // transpose of [[cos, -sin], [sin, cos]] (:355-364)
    const double wy = (-sa * ox + ca * oy) + py32;
    const bool is_left = k < a.k_left;
    const Track t = is_left ? a.left : a.right;
    const int i = is_left ? k : k - a.k_left;
    const int count = is_left ? a.k_left : a.k_right;
    const int closest = is_left ? i_left : i_right;
    // np.linspace(closest, closest + count, count, dtype=uint16): closest + i, except the last entry = closest + count
    const int off = (count > 1 && i == count - 1) ? count : i;
    const int idx = ((closest + off) & 0xffff) % t.m;
    const double dx = wx - t.xy[2 * idx], dy = wy - t.xy[2 * idx + 1];
    sum += sqrt(dx * dx + dy * dy);
  }
  s_d[tid] = sum;
  __syncthreads();
  for (int half = kBlock / 2; half > 0; half >>= 1) {  // fixed tree: reproducible
    if (tid < half) s_d[tid] += s_d[tid + half];
    __syncthreads();
  }

  if (tid == 0) {
    const double error = s_d[0] / static_cast<double>(K);
    // heading of the centreline at the nearest point, indices mod (len - 1) (:291-318)
    const int m1 = a.centre.m - 1;
    const int here = i_centre % m1, next = (i_centre + 1) % m1;
    const double track_heading = atan2(a.centre.xy[2 * next + 1] - a.centre.xy[2 * here + 1],
                                       a.centre.xy[2 * next] - a.centre.xy[2 * here]);
    const double raw = track_heading - phi + kPi;
    const double heading = fabs(raw - floor(raw / (2 * kPi)) * (2 * kPi) - kPi);
    const double offset = sqrt(d_centre);
    const double z = (error - a.mean) / a.sigma;
    const double score = exp(-z * z / 2.0) / sqrt(2.0 * kPi) / a.sigma / a.scale;
    a.track_indices[3 * p] = i_centre;
    a.track_indices[3 * p + 1] = i_left;
    a.track_indices[3 * p + 2] = i_right;
    a.minimum_offset[p] = offset;
    a.heading_offset[p] = heading;
    a.error[p] = error;
    a.score[p] = score;
    a.valid[p] = (heading < a.thr_rotation && offset < a.thr_offset && error < a.thr_error) ? 1 : 0;
  }
}

// states += x_dot * dt, x_dot = (v cos phi, v sin phi, v tan delta / L) in float32 as the reference computes it
// (localiser.py:66-95: float32 state array, per-particle delta and v)
__global__ void pf_advance_kernel(float* states, const float* delta, const float* velocity, int P, float wheelbase,
                                  float dt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float phi = states[3 * p + 2], v = velocity[p];
  states[3 * p] += (v * cosf(phi)) * dt;
  states[3 * p + 1] += (v * sinf(phi)) * dt;
  states[3 * p + 2] += (v * tanf(delta[p]) / wheelbase) * dt;
}

// sum(state * score) / sum(score) with the NaN -> uniform fallback, then max distance / max |yaw difference| to the
// estimate (localiser.py:561-579).  One workgroup; float64 accumulation in a fixed order.
__global__ void __launch_bounds__(kBlock) pf_estimate_kernel(const float* states, const float* scores, int P,
                                                             double* out /*[5]: x, y, yaw, max_dist, max_angle*/) {
  __shared__ double s[4][kBlock];
  const int tid = threadIdx.x;
  double acc[4] = {0, 0, 0, 0}, plain[3] = {0, 0, 0};
  for (int p = tid; p < P; p += kBlock) {
    const double w = scores[p];
    for (int c = 0; c < 3; ++c) {
      acc[c] += static_cast<double>(states[3 * p + c]) * w;
    }
    acc[3] += w;
  }
  for (int c = 0; c < 4; ++c) s[c][tid] = acc[c];
  __syncthreads();
  for (int half = kBlock / 2; half > 0; half >>= 1) {
    if (tid < half)
      for (int c = 0; c < 4; ++c) s[c][tid] += s[c][tid + half];
    __syncthreads();
  }
  double est[3] = {s[0][0] / s[3][0], s[1][0] / s[3][0], s[2][0] / s[3][0]};
  __syncthreads();
  if (est[0] != est[0] || est[1] != est[1] || est[2] != est[2]) {  // NaN: uniform weights
    for (int p = tid; p < P; p += kBlock)
      for (int c = 0; c < 3; ++c) plain[c] += static_cast<double>(states[3 * p + c]);
    for (int c = 0; c < 3; ++c) s[c][tid] = plain[c];
    __syncthreads();
    for (int half = kBlock / 2; half > 0; half >>= 1) {
      if (tid < half)
        for (int c = 0; c < 3; ++c) s[c][tid] += s[c][tid + half];
      __syncthreads();
    }
    for (int c = 0; c < 3; ++c) est[c] = s[c][0] / static_cast<double>(P);
    __syncthreads();
  }
  double md = 0.0, ma = 0.0;
  for (int p = tid; p < P; p += kBlock) {
    const double dx = states[3 * p] - est[0], dy = states[3 * p + 1] - est[1];
    md = fmax(md, sqrt(dx * dx + dy * dy));
    ma = fmax(ma, fabs(states[3 * p + 2] - est[2]));
  }
  s[0][tid] = md;
  s[1][tid] = ma;
  __syncthreads();
  for (int half = kBlock / 2; half > 0; half >>= 1) {
    if (tid < half) {
      s[0][tid] = fmax(s[0][tid], s[0][tid + half]);
      s[1][tid] = fmax(s[1][tid], s[1][tid + half]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = est[0];
    out[1] = est[1];
    out[2] = est[2];
    out[3] = s[0][0];
    out[4] = s[1][0];
  }
}

thread_local std::string g_pf_create_error;

}  // namespace

struct acmpc_pf {
  acmpc_pf_params prm{};
  std::vector<double> h_track[3];
  double scale = 1.0;
  bool device_ready = false;
  double* d_track[3] = {nullptr, nullptr, nullptr};
  hipStream_t stream = nullptr;
  // staging for the host-pointer entry points
  float* d_states = nullptr;
  float* d_obs = nullptr;
  float* d_aux = nullptr;      // 2 * max_particles floats: delta / velocity, or scores
  int32_t* d_indices = nullptr;
  double* d_out = nullptr;     // 4 * max_particles doubles + 8
  uint8_t* d_valid = nullptr;
  mutable std::string err;
};

namespace {

int pf_fail(const acmpc_pf* h, int code, const std::string& msg) {
  if (h != nullptr) {
    h->err = msg;
  } else {
    g_pf_create_error = msg;
  }
  return code;
}

#define PF_HIP(h, call)                                                                                   \
  do {                                                                                                    \
    const hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                                 \
      return pf_fail((h), (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? ACMPC_ENODEVICE : ACMPC_EHIP, \
                     std::string(#call) + ": " + hipGetErrorString(e_));                                  \
  } while (0)

int pf_ensure_device(acmpc_pf* h) {
  if (h->device_ready) return ACMPC_OK;
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return pf_fail(h, ACMPC_ENODEVICE, "no HIP device visible: particle scoring has no CPU fallback");
  if (h->prm.device >= 0) PF_HIP(h, hipSetDevice(h->prm.device));
  PF_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (int t = 0; t < 3; ++t) {
    PF_HIP(h, hipMalloc(&h->d_track[t], h->h_track[t].size() * sizeof(double)));
    PF_HIP(h, hipMemcpy(h->d_track[t], h->h_track[t].data(), h->h_track[t].size() * sizeof(double),
                        hipMemcpyHostToDevice));
  }
  const size_t P = h->prm.max_particles, K = h->prm.max_observation_points;
  PF_HIP(h, hipMalloc(&h->d_states, P * 3 * sizeof(float)));
  PF_HIP(h, hipMalloc(&h->d_obs, K * 2 * sizeof(float)));
  PF_HIP(h, hipMalloc(&h->d_aux, P * 2 * sizeof(float)));
  PF_HIP(h, hipMalloc(&h->d_indices, P * 3 * sizeof(int32_t)));
  PF_HIP(h, hipMalloc(&h->d_out, (P * 4 + 8) * sizeof(double)));
  PF_HIP(h, hipMalloc(&h->d_valid, P));
  h->device_ready = true;
  return ACMPC_OK;
}

}  // namespace

extern "C" {

const char* acmpc_pf_last_error(const acmpc_pf* h) { return h != nullptr ? h->err.c_str() : g_pf_create_error.c_str(); }

int acmpc_pf_create(const acmpc_pf_params* params, const double* centre, int32_t m_centre, const double* left,
                    int32_t m_left, const double* right, int32_t m_right, acmpc_pf** out) {
  if (params == nullptr || centre == nullptr || left == nullptr || right == nullptr || out == nullptr)
    return pf_fail(nullptr, ACMPC_EINVAL, "null argument");
  *out = nullptr;
  if (params->struct_size != sizeof(acmpc_pf_params)) return pf_fail(nullptr, ACMPC_EINVAL, "acmpc_pf_params size mismatch");
  if (m_centre < 3 || m_left < 1 || m_right < 1) return pf_fail(nullptr, ACMPC_EINVAL, "map polylines too short");
  if (params->max_particles < 1 || params->max_observation_points < 1 || !(params->score_sigma > 0.0))
    return pf_fail(nullptr, ACMPC_EINVAL, "bad capacities or score_sigma");
  acmpc_pf* h = new (std::nothrow) acmpc_pf();
  if (h == nullptr) return pf_fail(nullptr, ACMPC_EINVAL, "out of host memory");
  h->prm = *params;
  h->h_track[0].assign(centre, centre + 2 * static_cast<size_t>(m_centre));
  h->h_track[1].assign(left, left + 2 * static_cast<size_t>(m_left));
  h->h_track[2].assign(right, right + 2 * static_cast<size_t>(m_right));
  // score normaliser: max of the pdf over linspace(-10, 10, 100) (localiser.py:655-661)
  double best = 0.0;
  for (int i = 0; i < 100; ++i) {
    const double x = -10.0 + 20.0 * i / 99.0;
    const double z = (x - params->score_mean) / params->score_sigma;
    best = std::max(best, std::exp(-z * z / 2.0) / std::sqrt(2.0 * kPi) / params->score_sigma);
  }
  h->scale = best;
  *out = h;
  return ACMPC_OK;
}

void acmpc_pf_destroy(acmpc_pf* h) {
  if (h == nullptr) return;
  if (h->device_ready) {
    if (h->prm.device >= 0) (void)hipSetDevice(h->prm.device);
    for (int t = 0; t < 3; ++t) (void)hipFree(h->d_track[t]);
    (void)hipFree(h->d_states);
    (void)hipFree(h->d_obs);
    (void)hipFree(h->d_aux);
    (void)hipFree(h->d_indices);
    (void)hipFree(h->d_out);
    (void)hipFree(h->d_valid);
    if (h->stream != nullptr) (void)hipStreamDestroy(h->stream);
  }
  delete h;
}

double acmpc_pf_score_scale(const acmpc_pf* h) { return h != nullptr ? h->scale : 0.0; }

int acmpc_pf_score(acmpc_pf* h, const float* states, int32_t P, const float* obs_left, int32_t k_left,
                   const float* obs_right, int32_t k_right, int32_t* track_indices, double* minimum_offset,
                   double* heading_offset, double* observation_error, double* score, uint8_t* valid) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (states == nullptr || track_indices == nullptr || minimum_offset == nullptr || heading_offset == nullptr ||
      observation_error == nullptr || score == nullptr || valid == nullptr)
    return pf_fail(h, ACMPC_EINVAL, "null argument");
  if (P < 1 || k_left < 0 || k_right < 0 || k_left + k_right < 1) return pf_fail(h, ACMPC_EINVAL, "empty input");
  if ((k_left > 0 && obs_left == nullptr) || (k_right > 0 && obs_right == nullptr))
    return pf_fail(h, ACMPC_EINVAL, "null observation");
  if (P > h->prm.max_particles || k_left + k_right > h->prm.max_observation_points)
    return pf_fail(h, ACMPC_ECAPACITY, "more particles or observation points than the handle was created for");
  const int rc = pf_ensure_device(h);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = h->stream;
  PF_HIP(h, hipMemcpyAsync(h->d_states, states, static_cast<size_t>(P) * 3 * sizeof(float), hipMemcpyHostToDevice, s));
  if (k_left > 0)
    PF_HIP(h, hipMemcpyAsync(h->d_obs, obs_left, static_cast<size_t>(k_left) * 2 * sizeof(float), hipMemcpyHostToDevice, s));
  if (k_right > 0)
    PF_HIP(h, hipMemcpyAsync(h->d_obs + 2 * k_left, obs_right, static_cast<size_t>(k_right) * 2 * sizeof(float),
                             hipMemcpyHostToDevice, s));
  ScoreArgs a{};
  a.states = h->d_states;
  a.obs = h->d_obs;
  a.k_left = k_left;
  a.k_right = k_right;
  a.centre = Track{h->d_track[0], static_cast<int>(h->h_track[0].size() / 2)};
  a.left = Track{h->d_track[1], static_cast<int>(h->h_track[1].size() / 2)};
  a.right = Track{h->d_track[2], static_cast<int>(h->h_track[2].size() / 2)};
  a.mean = h->prm.score_mean;
  a.sigma = h->prm.score_sigma;
  a.scale = h->scale;
  a.thr_rotation = h->prm.threshold_rotation;
  a.thr_offset = h->prm.threshold_offset;
  a.thr_error = h->prm.threshold_error;
  a.track_indices = h->d_indices;
  a.minimum_offset = h->d_out;
  a.heading_offset = h->d_out + P;
  a.error = h->d_out + 2 * static_cast<size_t>(P);
  a.score = h->d_out + 3 * static_cast<size_t>(P);
  a.valid = h->d_valid;
  (void)hipGetLastError();  // a stale error of an earlier call must not be read as this launch's
  hipLaunchKernelGGL(pf_score_kernel, dim3(P), dim3(kBlock), 0, s, a);
  PF_HIP(h, hipGetLastError());
  const size_t pd = static_cast<size_t>(P) * sizeof(double);
  PF_HIP(h, hipMemcpyAsync(track_indices, h->d_indices, static_cast<size_t>(P) * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipMemcpyAsync(minimum_offset, a.minimum_offset, pd, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipMemcpyAsync(heading_offset, a.heading_offset, pd, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipMemcpyAsync(observation_error, a.error, pd, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipMemcpyAsync(score, a.score, pd, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipMemcpyAsync(valid, h->d_valid, static_cast<size_t>(P), hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipStreamSynchronize(s));
  return ACMPC_OK;
}

int acmpc_pf_advance(acmpc_pf* h, float* states, const float* delta, const float* velocity, int32_t P, double dt) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (states == nullptr || delta == nullptr || velocity == nullptr) return pf_fail(h, ACMPC_EINVAL, "null argument");
  if (P < 1 || P > h->prm.max_particles) return pf_fail(h, ACMPC_ECAPACITY, "particle count out of range");
  const int rc = pf_ensure_device(h);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = h->stream;
  const size_t pf = static_cast<size_t>(P) * sizeof(float);
  PF_HIP(h, hipMemcpyAsync(h->d_states, states, 3 * pf, hipMemcpyHostToDevice, s));
  PF_HIP(h, hipMemcpyAsync(h->d_aux, delta, pf, hipMemcpyHostToDevice, s));
  PF_HIP(h, hipMemcpyAsync(h->d_aux + P, velocity, pf, hipMemcpyHostToDevice, s));
  (void)hipGetLastError();  // a stale error of an earlier call must not be read as this launch's
  hipLaunchKernelGGL(pf_advance_kernel, dim3((P + 255) / 256), dim3(256), 0, s, h->d_states, h->d_aux, h->d_aux + P, P,
                     static_cast<float>(h->prm.wheelbase), static_cast<float>(dt));
  PF_HIP(h, hipGetLastError());
  PF_HIP(h, hipMemcpyAsync(states, h->d_states, 3 * pf, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipStreamSynchronize(s));
  return ACMPC_OK;
}

int acmpc_pf_estimate(acmpc_pf* h, const float* states, const float* scores, int32_t P, double estimate[3],
                      double* max_distance, double* max_angle) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (states == nullptr || scores == nullptr || estimate == nullptr) return pf_fail(h, ACMPC_EINVAL, "null argument");
  if (P < 1 || P > h->prm.max_particles) return pf_fail(h, ACMPC_ECAPACITY, "particle count out of range");
  const int rc = pf_ensure_device(h);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = h->stream;
  PF_HIP(h, hipMemcpyAsync(h->d_states, states, static_cast<size_t>(P) * 3 * sizeof(float), hipMemcpyHostToDevice, s));
  PF_HIP(h, hipMemcpyAsync(h->d_aux, scores, static_cast<size_t>(P) * sizeof(float), hipMemcpyHostToDevice, s));
  double* d_res = h->d_out + 4 * static_cast<size_t>(h->prm.max_particles);
  (void)hipGetLastError();  // a stale error of an earlier call must not be read as this launch's
  hipLaunchKernelGGL(pf_estimate_kernel, dim3(1), dim3(kBlock), 0, s, h->d_states, h->d_aux, P, d_res);
  PF_HIP(h, hipGetLastError());
  double res[5];
  PF_HIP(h, hipMemcpyAsync(res, d_res, sizeof res, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipStreamSynchronize(s));
  estimate[0] = res[0];
  estimate[1] = res[1];
  estimate[2] = res[2];
  if (max_distance != nullptr) *max_distance = res[3];
  if (max_angle != nullptr) *max_angle = res[4];
  return ACMPC_OK;
}

}  // extern "C"
