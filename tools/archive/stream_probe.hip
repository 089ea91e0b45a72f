// Development probe: what read bandwidth does this GPU give (a) a flat 16 B/lane streaming sum and (b) the
// rollout kernel's access pattern (step-major rows, CPT adjacent candidates per lane, n steps) with trivial
// arithmetic?  Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void flat_sum(const f32x4* __restrict__ in, float* __restrict__ out, size_t n4) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i < n4; i += stride) {
    const f32x4 v = in[i];
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int CPT, int UNROLL>
__global__ void __launch_bounds__(256) pattern_sum(const float* __restrict__ U, float* __restrict__ costs, int N, int n) {
  const int p = blockIdx.y;
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * CPT;
  if (c0 >= N) return;
  float acc[CPT] = {};
#pragma unroll UNROLL
  for (int i = 0; i < n; ++i) {
    const float* row = U + ((size_t)p * n + i) * 2 * (size_t)N + c0;
    if constexpr (CPT == 1) {
      acc[0] += row[0] * 0.5f + row[N];
    } else if constexpr (CPT == 2) {
      const f32x2 a = *reinterpret_cast<const f32x2*>(row), b = *reinterpret_cast<const f32x2*>(row + N);
      acc[0] += a[0] * 0.5f + b[0];
      acc[1] += a[1] * 0.5f + b[1];
    } else {
      const f32x4 a = *reinterpret_cast<const f32x4*>(row), b = *reinterpret_cast<const f32x4*>(row + N);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += a[j] * 0.5f + b[j];
    }
  }
#pragma unroll
  for (int j = 0; j < CPT; ++j) costs[(size_t)p * N + c0 + j] = acc[j];
}

// the rollout's pattern with its non-temporal loads, BLOCK threads x 4 adjacent candidates per lane, `work` dependent
// multiply-adds per step and candidate standing in for the model (0: trivial arithmetic)
template <int BLOCK, int UNROLL>
__global__ void __launch_bounds__(BLOCK) pattern_nt(const float* __restrict__ U, float* __restrict__ costs, int N, int n, int work) {
  const int p = blockIdx.y;
  const int c0 = (blockIdx.x * BLOCK + threadIdx.x) * 4;
  if (c0 >= N) return;
  float acc[4] = {};
#pragma unroll UNROLL
  for (int i = 0; i < n; ++i) {
    const float* row = U + ((size_t)p * n + i) * 2 * (size_t)N + c0;
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row));
    const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row + N));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = a[j] * 0.5f + b[j];
      for (int w = 0; w < work; ++w) x = x * 1.0001f + acc[j] * 0.25f;
      acc[j] += x;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) costs[(size_t)p * N + c0 + j] = acc[j];
}

template <typename F>
float time_us(F f, int iters = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  std::vector<float> t;
  for (int i = 0; i < iters + 3; ++i) {
    hipEventRecord(a);
    f(i);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (i >= 3) t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 256, N = 4096, n = 49;   // (4096: the headline's batch, 6.6 GB per matrix)
  const size_t floats = (size_t)P * n * 2 * N;
  float *U[2], *costs;
  for (int b = 0; b < 2; ++b) {
    hipMalloc(&U[b], floats * 4);
    hipMemset(U[b], 0x3c, floats * 4);
  }
  hipMalloc(&costs, (size_t)P * N * 4);
  const double bytes = floats * 4.0;
  for (int blocks : {2048, 8192, 32768}) {
    float us = time_us([&](int i) { flat_sum<<<blocks, 256>>>((const f32x4*)U[i & 1], costs, floats / 4); });
    printf("flat float4 sum   grid %5d x256 : %7.1f us  %.2f TB/s\n", blocks, us, bytes / us / 1e6);
  }
  {
    float us = time_us([&](int i) { pattern_sum<1, 7><<<dim3(N / 256, P), 256>>>(U[i & 1], costs, N, n); });
    printf("pattern CPT1 unroll7            : %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_sum<2, 7><<<dim3(N / 512, P), 256>>>(U[i & 1], costs, N, n); });
    printf("pattern CPT2 unroll7            : %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_sum<4, 7><<<dim3(N / 1024, P), 256>>>(U[i & 1], costs, N, n); });
    printf("pattern CPT4 unroll7            : %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_sum<2, 14><<<dim3(N / 512, P), 256>>>(U[i & 1], costs, N, n); });
    printf("pattern CPT2 unroll14           : %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_sum<1, 49><<<dim3(N / 256, P), 256>>>(U[i & 1], costs, N, n); });
    printf("pattern CPT1 unroll49           : %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_sum<4, 49><<<dim3(N / 1024, P), 256>>>(U[i & 1], costs, N, n); });
    printf("pattern CPT4 unroll49           : %7.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
  }
  for (int work : {0, 12}) {
    float us = time_us([&](int i) { pattern_nt<256, 7><<<dim3(N / 1024, P), 256>>>(U[i & 1], costs, N, n, work); });
    printf("nt pattern 256 x4, work %2d      : %7.1f us  %.2f TB/s\n", work, us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_nt<512, 7><<<dim3(N / 2048, P), 512>>>(U[i & 1], costs, N, n, work); });
    printf("nt pattern 512 x4, work %2d      : %7.1f us  %.2f TB/s\n", work, us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_nt<1024, 7><<<dim3(N / 4096, P), 1024>>>(U[i & 1], costs, N, n, work); });
    printf("nt pattern 1024 x4, work %2d     : %7.1f us  %.2f TB/s\n", work, us, bytes / us / 1e6);
    us = time_us([&](int i) { pattern_nt<128, 7><<<dim3(N / 512, P), 128>>>(U[i & 1], costs, N, n, work); });
    printf("nt pattern 128 x4, work %2d      : %7.1f us  %.2f TB/s\n", work, us, bytes / us / 1e6);
  }
  return 0;
}
