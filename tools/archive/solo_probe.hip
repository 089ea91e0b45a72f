// Development probe: where one launch of rollout_solo_kernel spends its time.  A standalone build of the library's
// kernel source with -DACMPC_STAMPS: lane 0 of every wave stamps the 100 MHz wall clock at the phase boundaries.
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DACMPC_STAMPS tools/solo_probe.hip -o /tmp/solo_probe
//   /tmp/solo_probe <N> <n> <layout> [split 0|1]
#include "../ac-mpc_amd/csrc/acmpc_kernels.hip"

#include <algorithm>
#include <random>
#include <vector>

// (the two launchers the library builds in its other translation unit: not used here)
namespace acmpc {
hipError_t launch_rollout_tile_rows_plain(const LaunchShape&, const RolloutArgs&, hipStream_t, hipEvent_t, hipEvent_t) {
  return hipErrorNotSupported;
}
hipError_t launch_rollout_temporal_plain(const LaunchShape&, const RolloutArgs&, hipStream_t, hipEvent_t, hipEvent_t) {
  return hipErrorNotSupported;
}
}  // namespace acmpc

#define CHECK(call)                                                                          \
  do {                                                                                       \
    const hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                        \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 4096;
  const int n = argc > 2 ? std::atoi(argv[2]) : 49;
  const int layout = argc > 3 ? std::atoi(argv[3]) : 1;
  if (argc > 4) setenv("ACMPC_SOLO_SPLIT", argv[4], 1);
  const int blocks = (N + 63) / 64;
  std::mt19937 rng(1);
  std::normal_distribution<float> nv(20.0f, 2.0f), nk(0.0f, 0.01f);
  std::vector<float> U(static_cast<size_t>(N) * n * 2), coef(static_cast<size_t>(n) * 12, 0.0f), x0 = {0.1f, 0.0f, 0.0f};
  for (size_t e = 0; e < U.size(); e += 2) {
    U[e] = nv(rng);
    U[e + 1] = nk(rng);
  }
  if (layout == 1) {   // (the values need not correspond between layouts)
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < N; ++c) {
        U[(static_cast<size_t>(i) * 2) * N + c] = nv(rng);
        U[(static_cast<size_t>(i) * 2 + 1) * N + c] = nk(rng);
      }
  }
  for (int i = 0; i < n; ++i) {
    float* r = coef.data() + i * 12;
    r[0] = 3.0f; r[1] = -1e-4f; r[2] = -1e-3f; r[3] = -1e-3f; r[4] = 0.15f; r[5] = 20.0f; r[6] = 0.0f; r[7] = -3.0f; r[8] = 3.0f;
  }
  float *d_U, *d_coef, *d_x0, *d_costs, *d_rec, *d_trace;
  int64_t *d_keys, *d_pk;
  int *d_pf, *d_tickets;
  CHECK(hipMalloc(&d_U, U.size() * 4));
  CHECK(hipMalloc(&d_coef, coef.size() * 4));
  CHECK(hipMalloc(&d_x0, 12));
  CHECK(hipMalloc(&d_costs, static_cast<size_t>(N) * 4));
  CHECK(hipMalloc(&d_rec, (4 + 5 * n + 3) * 4));
  CHECK(hipMalloc(&d_trace, static_cast<size_t>(blocks) * (3 * n + 2) * 4));
  CHECK(hipMalloc(&d_keys, 8));
  CHECK(hipMalloc(&d_pk, static_cast<size_t>(blocks) * 8));
  CHECK(hipMalloc(&d_pf, static_cast<size_t>(blocks) * 4));
  CHECK(hipMalloc(&d_tickets, 33 * 4));
  CHECK(hipMemset(d_tickets, 0, 33 * 4));
  CHECK(hipMemcpy(d_U, U.data(), U.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_coef, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_x0, x0.data(), 12, hipMemcpyHostToDevice));
  acmpc::RolloutArgs a{};
  a.U = d_U; a.x0 = d_x0; a.coef = d_coef; a.costs = d_costs; a.partial_keys = d_pk; a.partial_feas = d_pf;
  a.P = 1; a.N = N; a.n = n;
  a.w.q0 = 1.0f; a.w.q1 = 0.1f; a.w.q2 = 0.01f; a.w.r0 = 0.01f; a.w.r1 = 1.0f; a.w.qn0 = 1.0f; a.w.qn1 = 0.1f; a.w.qn2 = 0.01f;
  a.w.ulo0 = 0.0f; a.w.uhi0 = 60.0f; a.w.ulo1 = -0.1f; a.w.uhi1 = 0.1f; a.w.tmin = 0.01f; a.w.wbound = 1e6f;
  acmpc::FusedFinalize ff{};
  ff.tickets = d_tickets; ff.records = d_rec; ff.trace = d_trace; ff.trace_pitch = 3 * n + 2; ff.keys_out = d_keys;
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < 30; ++it) {
    CHECK(acmpc::launch_rollout_solo(layout, a, ff, s, e0, e1));
    CHECK(hipStreamSynchronize(s));
    float t;
    CHECK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  std::printf("N=%d n=%d layout=%d blocks=%d: kernel median %.2f us (min %.2f)\n", N, n, layout, blocks, ms[15] * 1e3, ms[0] * 1e3);
  std::vector<unsigned long long> st(4096 * 16);
  CHECK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(acmpc::g_stamps), st.size() * 8));
  // per wave: stamps relative to the earliest stamp 0 of the launch, in us (100 MHz clock)
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < blocks; ++b) t0 = std::min(t0, st[(b * 2) * 16]);
  auto us = [&](unsigned long long v) { return (static_cast<double>(v) - static_cast<double>(t0)) * 0.01; };
  const char* names[10] = {"start", "loop begins", "loop ends", "reduced", "published (issued)", "stores acknowledged",
                           "group ticket", "problem ticket", "keys in", "record written"};
  // wave 0 of the first and the last-starting workgroup, and whoever wrote stamp 9 in this launch (the finalizer)
  int finalizer = -1;
  unsigned long long newest = 0;
  for (int b = 0; b < blocks; ++b)
    if (st[(b * 2) * 16 + 9] > newest) { newest = st[(b * 2) * 16 + 9]; finalizer = b; }
  int last_start = 0;
  for (int b = 0; b < blocks; ++b)
    if (st[(b * 2) * 16] > st[(last_start * 2) * 16]) last_start = b;
  for (int b : {0, last_start, finalizer}) {
    std::printf("workgroup %d wave 0:", b);
    for (int k = 0; k < 10; ++k)
      if (k < 6 || b == finalizer) std::printf("  %s %.2f", names[k], us(st[(b * 2) * 16 + k]));
    std::printf("\n    wave 1: start %.2f loop %.2f .. %.2f\n", us(st[(b * 2 + 1) * 16]), us(st[(b * 2 + 1) * 16 + 1]),
                us(st[(b * 2 + 1) * 16 + 2]));
  }
  // spread of the phase ends over all workgroups
  for (int k : {0, 2, 3, 5}) {
    std::vector<double> v;
    for (int b = 0; b < blocks; ++b) v.push_back(us(st[(b * 2) * 16 + k]));
    std::sort(v.begin(), v.end());
    std::printf("%-20s over workgroups: min %.2f median %.2f max %.2f\n", names[k], v.front(), v[v.size() / 2], v.back());
  }
  return 0;
}
