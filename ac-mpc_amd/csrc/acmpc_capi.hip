// C ABI of the rollout-and-cost engine (include/acmpc.h): handle, host-side table preparation, lazy device
// bring-up, and the launch sequences.  No CPU fallback exists: every compute entry point needs the GPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: RCCL is resolved at run time, not linked

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/acmpc.h"
#include "acmpc_frames.h"
#include "acmpc_kernels.h"
#include "acmpc_lq.h"
#include "acmpc_lq_box.h"
#include "acmpc_prologue.h"

namespace {

thread_local std::string g_create_error;

constexpr double kEps = 1e-12;  // dynamics.py:21
constexpr int kTraceBlocks = 1024;  // workgroups per launch the traced fused finalize has room for (64 candidates each)

}  // namespace

struct acmpc_ctx {
  acmpc_params prm{};
  acmpc::Weights w{};
  int coef_stride = 0;

  // host copy of the packed tables
  std::vector<float> h_coef;
  int P_set = 0, n_set = 0;
  bool tables_dirty = false;
  bool frames_dirty = false;  // the verified search's frames of the current paths are not on the device yet

  // device state (created lazily)
  bool device_ready = false;
  bool touched_device = false;  // a HIP call has been made for this handle (acmpc_destroy must not make the first one)
  float* d_coef = nullptr;
  int64_t* d_partial_keys = nullptr;
  int* d_partial_feas = nullptr;
  size_t partial_slots = 0;  // slots of ONE set of partial keys / feasible counts (there are two)
  // acmpc_solve_stream_device: the finalize the last call of the stream left for the next one (or for the flush)
  bool stream_pending = false;
  acmpc::FinalizeArgs stream_fin{};
  int stream_fin_layout = 0;
  int stream_set = 0;        // the half of the partial buffers the pending finalize reads
  double* d_soft_partial = nullptr;
  size_t soft_partial_doubles = 0;

  // sampler: per-step (left knot, weight) table, uploaded when n changes
  float* d_segments = nullptr;
  int segments_n = 0;
  int knot_begin[acmpc::kKnots + 1] = {};
  float* d_centre = nullptr;  // [P][n][2] staging of acmpc_optimize (first round's centre, then u_ref)
  float* d_uref = nullptr;

  // staging for the host-pointer entry point (created on its first use)
  bool staging_ready = false;
  hipStream_t stream = nullptr;
  float* d_U = nullptr;
  float* d_x0 = nullptr;
  float* d_costs = nullptr;
  float* d_records = nullptr;
  int64_t* d_keys = nullptr;
  int* d_tickets = nullptr;  // last-workgroup counters of the in-launch finalizes (ensure_tail_buffers); zero between launches
  unsigned tick_sequence = 0;   // completion flag values of acmpc_control_tick
  float* d_trace = nullptr;  // [2][kTraceBlocks][trace_floats(max_steps)] best-candidate traces of the fused rounds' workgroups
  // mode T with exhaustive search: frames of the verified window search (acmpc_device.h: nearest_verified)
  std::vector<float> h_nn_frames;  // [P][verified_frame_floats(n)], empty when not applicable
  float* d_nn_frames = nullptr;
  int64_t* h_keys = nullptr;  // pinned
  float* h_io = nullptr;      // pinned: x0 [P][3] on the way up, records [P][record_floats] on the way down (acmpc_solve)

  // acmpc_optimize as a hipGraph: the whole sample -> rollout -> finalize chain of `rounds` rounds plus the
  // transfers either side of it is captured once per shape and replayed; per-call inputs travel through the pinned
  // staging block `h_opt` (x0 | centre | u_ref | table | seed) and the records come back into `h_opt_records`.
  struct OptKey {
    int P = 0, N = 0, n = 0, rounds = 0, has_uref = 0;
    double sigma_v = 0, sigma_k = 0, shrink = 0;
    bool operator==(const OptKey& o) const {
      return P == o.P && N == o.N && n == o.n && rounds == o.rounds && has_uref == o.has_uref &&
             sigma_v == o.sigma_v && sigma_k == o.sigma_k && shrink == o.shrink;
    }
  };
  // a few captured shapes side by side (a controller alternates between its exploring and its refining schedule);
  // the least recently used slot is re-captured when a new shape arrives
  static constexpr int kOptGraphs = 4;
  hipGraphExec_t opt_graph[kOptGraphs] = {nullptr, nullptr, nullptr, nullptr};
  OptKey opt_key[kOptGraphs];
  uint64_t opt_used[kOptGraphs] = {0, 0, 0, 0};
  uint64_t opt_clock = 0;
  bool opt_ready = false;
  unsigned char* h_opt = nullptr;   // pinned
  unsigned char* d_opt = nullptr;   // device mirror of h_opt: ONE H2D copy per solve
  size_t opt_capacity = 0;
  float* h_opt_records = nullptr;   // pinned
  uint32_t* d_seed = nullptr;

  // acmpc_control_tick: prologue + rounds as one captured graph per (N, n, rounds, spread); per-tick inputs travel
  // through the pinned block `h_tick` (TickHeader | coords | centre), results come back into `h_tick_out`
  // (record | table | QP status) by posted writes
  struct TickKey {
    int N = 0, n = 0, rounds = 0, from_map = 0;
    double sigma_v = 0, sigma_k = 0, shrink = 0;
    bool operator==(const TickKey& o) const {
      return N == o.N && n == o.n && rounds == o.rounds && from_map == o.from_map && sigma_v == o.sigma_v &&
             sigma_k == o.sigma_k && shrink == o.shrink;
    }
  };
  bool tick_ready = false;
  hipGraphExec_t tick_graph[kOptGraphs] = {nullptr, nullptr, nullptr, nullptr};
  TickKey tick_key[kOptGraphs];
  uint64_t tick_used[kOptGraphs] = {0, 0, 0, 0};
  unsigned char* h_tick = nullptr;      // pinned
  unsigned char* d_tick = nullptr;
  unsigned char* h_tick_out = nullptr;  // pinned
  std::vector<double> h_map;            // bound map: centre polyline [M][2]
  double map_spacing = 0.0;
  bool map_dirty = false;
  double* d_map = nullptr;
  double* d_coords = nullptr;           // [H][3] path the window kernel builds for the prologue
  double* d_warm = nullptr;             // speed-profile iterate of the two solvers, kept between ticks
  int warm_stride = 0;
  int tick_last_n = 0;

  // the LQ plan (csrc/acmpc_lq.h; acmpc_params::lq_candidate): candidate 2 of the LAST sampling round
  std::vector<double> h_tables;         // the float64 tables of acmpc_set_paths: [P][7][n]
  float* h_lq = nullptr;                // pinned [max_problems][max_steps][2]: the plans, read by the last round in place
  std::vector<double> tick_prev_table;  // what the previous acmpc_control_tick solved: its 7 x n table ...
  double tick_prev_x0[3] = {0.0, 0.0, 0.0};   // ... and its start state (Frenet)
  std::vector<double> tick_lq_table;    // scratch: this tick's waypoints with the speed profile the host plans with
  std::vector<double> tick_lq_scratch;  // scratch: its ceiling and (unused) multipliers
  std::vector<double> tick_host_coords; // scratch: the H x 3 path of a map window, cut on the host for the plan
  int tick_prev_n = 0;                  // 0: nothing usable (first tick, or a tick that did not end with a finite plan)
  // lq_candidate = 2 (csrc/acmpc_lq_box.h): the splitting's iterate per problem, its factorisation scratch, what the last
  // plan did (acmpc_lq_box_stats) and the iteration cap (ACMPC_LQ_BOX_ITERATIONS)
  // ACMPC_START_CLOCKS: the rollout launches leave every workgroup's start time here (acmpc_rollout_start_clocks)
  bool want_start_clocks = false;
  unsigned long long* d_start_clock = nullptr;
  size_t start_clock_slots = 0;
  int start_clock_count = 0;
  std::vector<acmpc::lqbox::State> lq_box_state;
  acmpc::lqbox::Workspace lq_box_ws;
  acmpc::lqbox::Result lq_box_last;
  int lq_box_iterations = 40;

  // A/B switches of the tests and the tools: read from the environment ONCE, by acmpc_create, or set with acmpc_set_option;
  // nothing on a launch path calls getenv
  acmpc::LaunchOptions opt;
  struct Switches {
    bool no_verified_search = false, no_solo = false, no_fused_finalize = false, no_traced_finalize = false,
         no_chained_rounds = false, no_chained_stream = false, no_graph = false, no_fused_sampling = false, tick_graph = false, tick_no_flag = false, tick_no_inline_path = false, no_zero_copy = false,
         tailed_rollout = false;
  } sw;

  // optional timing of the rollout dispatches (acmpc_profile_*): event pairs attached to the launches
  std::vector<hipEvent_t> prof_start, prof_stop;
  size_t prof_used = 0;

  mutable std::string err;
};

namespace {

int fail(const acmpc_ctx* ctx, int code, const std::string& msg) {
  if (ctx != nullptr) {
    ctx->err = msg;
  } else {
    g_create_error = msg;
  }
  return code;
}

int fail_hip(const acmpc_ctx* ctx, hipError_t e, const char* what) {
  (void)hipGetLastError();  // do not leave the error behind for the next launch check (of this or any other library)
  const bool nodev = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver ||
                      e == hipErrorNotInitialized);
  return fail(ctx, nodev ? ACMPC_ENODEVICE : ACMPC_EHIP,
              std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")");
}

#define ACMPC_HIP(ctx, call)                                   \
  do {                                                         \
    const hipError_t e_ = (call);                              \
    if (e_ != hipSuccess) return fail_hip((ctx), e_, #call);   \
  } while (0)

// Frames of mode T's verified nearest-waypoint search for P paths of n packed waypoint rows (acmpc_frames.h has the
// arithmetic and why it is sound).  O(n^2) per path, hence the cap on n.
constexpr int kMaxVerifiedSteps = 256;

void verified_frames(const float* coef, int P, int n, std::vector<float>* out) {
  constexpr int W = acmpc::kVerifiedWindow;
  const int windows = n - W + 1;
  const int floats = acmpc::verified_frame_floats(n);
  out->assign(static_cast<size_t>(P) * floats, 0.0f);
  std::vector<acmpc::frames::Geometry> geometry(static_cast<size_t>(windows));
  for (int p = 0; p < P; ++p) {
    const float* t = coef + static_cast<size_t>(p) * n * acmpc::kCoefT;
    // positions in the path's own frame, as the kernels take them (acmpc_device.h: start_temporal): float32 differences
    const float ox = t[0], oy = t[1];
    auto at = [t, ox, oy](int m, double& x, double& y) {
      x = static_cast<double>(t[m * acmpc::kCoefT] - ox);
      y = static_cast<double>(t[m * acmpc::kCoefT + 1] - oy);
    };
    double wn = 0.0;   // largest norm of a waypoint
    bool finite = true;
    for (int m = 0; m < n; ++m) {
      double x, y;
      at(m, x, y);
      finite = finite && std::isfinite(x) && std::isfinite(y);
      wn = std::max(wn, std::sqrt(x * x + y * y));
    }
    float largest_gap2 = 0.0f;
    for (int m = 0; m + 1 < n; ++m) {
      const float gap2 = acmpc::frames::squared_gap(t[(m + 1) * acmpc::kCoefT] - ox, t[(m + 1) * acmpc::kCoefT + 1] - oy,
                                                    t[m * acmpc::kCoefT] - ox, t[m * acmpc::kCoefT + 1] - oy);
      largest_gap2 = (gap2 > largest_gap2) ? gap2 : largest_gap2;
    }
    const int first = acmpc::frames::near_first(largest_gap2);
    double slab_max = 0.0;
    for (int lo = 0; lo < windows; ++lo) {
      double R = finite ? acmpc::frames::far_distance(at, n, lo, first) : 0.0;
      const int near = finite ? acmpc::frames::choose_near(at, n, lo, first, R) : first;
      geometry[lo] = acmpc::frames::window_geometry<0>(at, n, lo, finite, near, R);
      if (geometry[lo].usable) slab_max = std::max(slab_max, geometry[lo].aA - geometry[lo].aB);
    }
    const acmpc::frames::Scale scale = acmpc::frames::path_scale(wn, slab_max);
    float* table = out->data() + static_cast<size_t>(p) * floats;
    for (int lo = 0; lo < windows; ++lo) acmpc::frames::frame_row(geometry[lo], scale, table + acmpc::kFrameStride * lo);
  }
}

// Allocate only what is not there yet: after a mid-way failure (out of memory) the buffers already obtained stay
// owned by the handle, a retry on the same handle picks up where the failed call stopped, and acmpc_destroy frees
// whatever exists.
template <typename T>
hipError_t alloc_once(T** slot, size_t bytes) {
  if (*slot != nullptr) return hipSuccess;
  return hipMalloc(reinterpret_cast<void**>(slot), bytes);
}
template <typename T>
hipError_t host_alloc_once(T** slot, size_t bytes) {
  if (*slot != nullptr) return hipSuccess;
  return hipHostMalloc(reinterpret_cast<void**>(slot), bytes, hipHostMallocDefault);
}

int ensure_device(acmpc_ctx* c) {
  if (c->device_ready) return ACMPC_OK;
  c->touched_device = true;
  int count = 0;
  const hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    return fail(c, ACMPC_ENODEVICE,
                "no HIP device visible: the rollout path has no CPU fallback (hipGetDeviceCount: " +
                    std::string(e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")");
  }
  if (c->prm.device >= 0) ACMPC_HIP(c, hipSetDevice(c->prm.device));
  const acmpc_params& p = c->prm;
  const size_t coef_floats = static_cast<size_t>(p.max_problems) * p.max_steps * c->coef_stride;
  const size_t partials = static_cast<size_t>(p.max_problems) * acmpc::max_blocks_per_problem(p.max_candidates);
  c->soft_partial_doubles = static_cast<size_t>(p.max_problems) * acmpc::softmin_chunks(p.max_candidates) *
                            (4 * static_cast<size_t>(p.max_steps) + 1);
  ACMPC_HIP(c, alloc_once(&c->d_coef, coef_floats * sizeof(float)));
  // (two sets: chained optimisation rounds alternate, a round reads the keys its predecessor wrote while it writes its own)
  ACMPC_HIP(c, alloc_once(&c->d_partial_keys, 2 * partials * sizeof(int64_t)));
  ACMPC_HIP(c, alloc_once(&c->d_partial_feas, 2 * partials * sizeof(int)));
  c->partial_slots = partials;
  ACMPC_HIP(c, alloc_once(&c->d_soft_partial, c->soft_partial_doubles * sizeof(double)));
  ACMPC_HIP(c, alloc_once(&c->d_segments, static_cast<size_t>(p.max_steps) * 2 * sizeof(float)));
  if (p.mode == ACMPC_MODE_TEMPORAL && p.nn_ahead < 0)
    ACMPC_HIP(c, alloc_once(&c->d_nn_frames, static_cast<size_t>(p.max_problems) * sizeof(float) *
                                                 acmpc::verified_frame_floats(std::max(std::min(p.max_steps, kMaxVerifiedSteps),
                                                                                       acmpc::kVerifiedWindow))));
  c->device_ready = true;
  return ACMPC_OK;
}

int upload_tables(acmpc_ctx* c, hipStream_t s) {
  if (c->P_set == 0) return fail(c, ACMPC_ESTATE, "acmpc_set_paths has not been called");
  // pageable source: hipMemcpyAsync stages it before returning, so the host vectors may change afterwards.  (Round 4 tried
  // a page-locked staging block for small tables - a memcpy and a true asynchronous packet: 0.7 us of a 91 us
  // set_paths + solve, not worth the bookkeeping of when the block is free again.)
  if (c->frames_dirty && !c->h_nn_frames.empty() && c->d_nn_frames != nullptr) {
    ACMPC_HIP(c, hipMemcpyAsync(c->d_nn_frames, c->h_nn_frames.data(), c->h_nn_frames.size() * sizeof(float),
                                hipMemcpyHostToDevice, s));
    c->frames_dirty = false;
  }
  if (!c->tables_dirty) return ACMPC_OK;
  const size_t bytes = static_cast<size_t>(c->P_set) * c->n_set * c->coef_stride * sizeof(float);
  ACMPC_HIP(c, hipMemcpyAsync(c->d_coef, c->h_coef.data(), bytes, hipMemcpyHostToDevice, s));
  c->tables_dirty = false;
  return ACMPC_OK;
}

// The captured optimisation carries the coefficient table in its staging block but not the frames of mode T's
// verified nearest-waypoint search, which the three-kernel form of a round reads (rollout(): a.nn_frames): bring them
// up to date on the launch stream before the graph runs.
int upload_frames(acmpc_ctx* c, hipStream_t s) {
  if (!c->frames_dirty || c->h_nn_frames.empty() || c->d_nn_frames == nullptr) return ACMPC_OK;
  ACMPC_HIP(c, hipMemcpyAsync(c->d_nn_frames, c->h_nn_frames.data(), c->h_nn_frames.size() * sizeof(float),
                              hipMemcpyHostToDevice, s));
  c->frames_dirty = false;
  return ACMPC_OK;
}

int check_shape(acmpc_ctx* c, int P, int N, int n, int layout, bool stream_call = false) {
  if (c->stream_pending && !stream_call)
    return fail(c, ACMPC_ESTATE, "a batch of acmpc_solve_stream_device is pending: acmpc_solve_stream_flush first");
  if (P < 1 || N < 1 || n < 1) return fail(c, ACMPC_EINVAL, "P, N and n must be positive");
  if (layout != ACMPC_LAYOUT_CANDIDATE_MAJOR && layout != ACMPC_LAYOUT_STEP_MAJOR)
    return fail(c, ACMPC_EINVAL, "unknown layout");
  if (P > c->prm.max_problems || N > c->prm.max_candidates || n > c->prm.max_steps) {
    char buf[160];
    std::snprintf(buf, sizeof buf, "shape (P=%d, N=%d, n=%d) exceeds the handle's capacity (%d, %d, %d)", P, N, n,
                  c->prm.max_problems, c->prm.max_candidates, c->prm.max_steps);
    return fail(c, ACMPC_ECAPACITY, buf);
  }
  if (c->P_set == 0) return fail(c, ACMPC_ESTATE, "acmpc_set_paths has not been called");
  if (P != c->P_set || n != c->n_set) {
    char buf[160];
    std::snprintf(buf, sizeof buf, "shape (P=%d, n=%d) does not match the tables set (P=%d, n=%d)", P, n, c->P_set,
                  c->n_set);
    return fail(c, ACMPC_EINVAL, buf);
  }
  return ACMPC_OK;
}

int rollout(acmpc_ctx* c, const float* d_x0, const float* d_U, int P, int N, int n, int layout, int64_t offset,
            float* d_costs, hipStream_t s, acmpc::LaunchShape* shape_out) {
  const acmpc::LaunchShape shape = acmpc::choose_shape(P, N, layout, c->prm.mode, n, c->opt);
  acmpc::RolloutArgs a{};
  a.U = d_U;
  a.x0 = d_x0;
  a.coef = c->d_coef;
  a.nn_frames = (!c->h_nn_frames.empty() && !c->sw.no_verified_search) ? c->d_nn_frames : nullptr;
  a.costs = d_costs;
  a.partial_keys = c->d_partial_keys;
  a.partial_feas = c->d_partial_feas;
  a.P = P;
  a.N = N;
  a.n = n;
  a.index_offset = offset;
  a.w = c->w;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof_used < c->prof_start.size()) {
    e0 = c->prof_start[c->prof_used];
    e1 = c->prof_stop[c->prof_used];
    ++c->prof_used;
  }
  c->start_clock_count = 0;
  if (c->want_start_clocks && !shape.tile) {
    const size_t slots = static_cast<size_t>(P) * shape.blocks_per_problem;
    if (slots > c->start_clock_slots) {
      if (c->d_start_clock != nullptr) (void)hipFree(c->d_start_clock);
      c->d_start_clock = nullptr;
      c->start_clock_slots = 0;
      ACMPC_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_start_clock), slots * sizeof(unsigned long long)));
      c->start_clock_slots = slots;
    }
    a.start_clock = c->d_start_clock;
    c->start_clock_count = static_cast<int>(slots);
  }
  ACMPC_HIP(c, acmpc::launch_rollout(c->prm.mode, layout, shape, a, s, e0, e1));
  *shape_out = shape;
  return ACMPC_OK;
}

struct Regenerate {
  const float* d_centre;
  int centre_stride;
  const float* d_uref;
  acmpc::SampleSpec spec;
  const float* d_extra = nullptr;   // candidate 2's controls (the LQ plan), or nullptr
};

int finalize(acmpc_ctx* c, const int64_t* d_keys_in, int64_t* d_keys_out, const float* d_x0, const float* d_U, int P,
             int N, int n, int layout, int64_t offset, float* d_records, int blocks_per_problem, hipStream_t s,
             const Regenerate* regen = nullptr, const float* d_coef_override = nullptr) {
  acmpc::FinalizeArgs a{};
  if (regen != nullptr) {
    a.regenerate = true;
    a.centre = regen->d_centre;
    a.centre_stride = regen->centre_stride;
    a.u_ref = regen->d_uref;
    a.u_extra = regen->d_extra;
    a.spec = regen->spec;
  }
  a.U = d_U;
  a.x0 = d_x0;
  a.coef = d_coef_override != nullptr ? d_coef_override : c->d_coef;
  a.partial_keys = c->d_partial_keys;
  a.partial_feas = c->d_partial_feas;
  a.keys_in = d_keys_in;
  a.keys_out = d_keys_out;
  a.records = d_records;
  a.blocks_per_problem = blocks_per_problem;
  a.P = P;
  a.N = N;
  a.n = n;
  a.index_offset = offset;
  a.w = c->w;
  ACMPC_HIP(c, acmpc::launch_finalize(c->prm.mode, layout, a, s, c->opt));
  return ACMPC_OK;
}

// raised-cosine blend between kSampleKnots knots spread evenly over the n steps
int upload_segments(acmpc_ctx* c, int n, hipStream_t s) {
  if (c->segments_n == n) return ACMPC_OK;
  // a pending batch of acmpc_solve_stream_device re-draws its winners with the knot table of ITS horizon, in place in
  // d_segments: it must have run before the table is rewritten for another (acmpc_solve_stream_device flushes it itself)
  if (c->stream_pending)
    return fail(c, ACMPC_ESTATE, "a batch of acmpc_solve_stream_device with another horizon is pending: acmpc_solve_stream_flush first");
  std::vector<float> seg(static_cast<size_t>(n) * 2);
  const double width = static_cast<double>(n - 1) / (acmpc::kSampleKnots - 1);
  for (int i = 0; i < n; ++i) {
    const double pos = (n > 1) ? i / width : 0.0;
    int k0 = static_cast<int>(std::floor(pos));
    if (k0 > acmpc::kSampleKnots - 2) k0 = acmpc::kSampleKnots - 2;
    const double frac = pos - k0;
    seg[2 * i] = static_cast<float>(k0);
    seg[2 * i + 1] = static_cast<float>(0.5 * (1.0 + std::cos(3.14159265358979323846 * frac)));
  }
  // first step of every knot's segment (left knots are non-decreasing in the step index)
  for (int k = 0; k <= acmpc::kSampleKnots; ++k) c->knot_begin[k] = n;
  for (int i = n - 1; i >= 0; --i) c->knot_begin[static_cast<int>(seg[2 * i])] = i;
  for (int k = acmpc::kSampleKnots - 1; k >= 0; --k)
    if (c->knot_begin[k] > c->knot_begin[k + 1]) c->knot_begin[k] = c->knot_begin[k + 1];
  c->knot_begin[0] = 0;
  ACMPC_HIP(c, hipMemcpyAsync(c->d_segments, seg.data(), seg.size() * sizeof(float), hipMemcpyHostToDevice, s));
  ACMPC_HIP(c, hipStreamSynchronize(s));  // `seg` is a local
  c->segments_n = n;
  return ACMPC_OK;
}

acmpc::SampleSpec make_spec(const acmpc_ctx* c, double sigma_v, double sigma_k, uint64_t seed, uint32_t round) {
  acmpc::SampleSpec sp{};
  sp.segments = c->d_segments;
  sp.seed_lo = static_cast<uint32_t>(seed);
  sp.seed_hi = static_cast<uint32_t>(seed >> 32);
  sp.seed_ptr = nullptr;
  for (int k = 0; k <= acmpc::kKnots; ++k) sp.knot_begin[k] = c->knot_begin[k];
  sp.round = round;
  sp.sigma_v = static_cast<float>(sigma_v);
  sp.sigma_k = static_cast<float>(sigma_k);
  sp.ulo0 = c->w.ulo0;
  sp.ulo1 = c->w.ulo1;
  sp.uhi0 = c->w.uhi0;
  sp.uhi1 = c->w.uhi1;
  return sp;
}

int sample(acmpc_ctx* c, const float* d_centre, int centre_stride, const float* d_uref, int P, int N, int n,
           int layout, int64_t offset, double sigma_v, double sigma_k, uint64_t seed, uint32_t round, float* d_U,
           hipStream_t s, const uint32_t* d_seed = nullptr, const float* d_extra = nullptr) {
  const int rc = upload_segments(c, n, s);  // no-op once the table for this n is resident
  if (rc != ACMPC_OK) return rc;
  acmpc::SampleArgs a{};
  a.centre = d_centre;
  a.u_ref = d_uref;
  a.u_extra = d_extra;
  a.U = d_U;
  a.centre_stride = centre_stride;
  a.P = P;
  a.N = N;
  a.n = n;
  a.index_offset = offset;
  a.spec = make_spec(c, sigma_v, sigma_k, seed, round);
  a.spec.seed_ptr = d_seed;
  ACMPC_HIP(c, acmpc::launch_sample(layout, a, s));
  return ACMPC_OK;
}

// what the in-launch finalize of the fused rounds and of the one-launch solve needs: ticket counters (zero between
// launches) and the workgroups' traces
int ensure_tail_buffers(acmpc_ctx* c) {
  if (c->d_tickets != nullptr && c->d_trace != nullptr) return ACMPC_OK;
  c->touched_device = true;
  const acmpc_params& p = c->prm;
  // [P][kTicketGroups + 1] for the fused rounds, [P][up to kTicketGroupsMax + 1] for the one-launch solve (which takes at
  // most kSoloBlocks workgroups, so at most that many problems), every counter on a line of its own
  const size_t ticket_ints =
      std::max(static_cast<size_t>(p.max_problems) * (acmpc::kTicketGroups + 1),
               static_cast<size_t>(std::min(p.max_problems, acmpc::kSoloBlocks)) * (acmpc::kTicketGroupsMax + 1)) *
      acmpc::kTicketStride;
  if (c->d_tickets == nullptr) {
    ACMPC_HIP(c, alloc_once(&c->d_tickets, ticket_ints * sizeof(int)));
    ACMPC_HIP(c, hipMemset(c->d_tickets, 0, ticket_ints * sizeof(int)));
    ACMPC_HIP(c, hipStreamSynchronize(nullptr));  // the callers' streams do not order against the null stream
  }
  ACMPC_HIP(c, alloc_once(&c->d_trace, 2 * static_cast<size_t>(kTraceBlocks) * acmpc::trace_floats(p.max_steps) * sizeof(float)));
  return ACMPC_OK;
}

// acmpc_solve_device / acmpc_solve in ONE launch (rollout_solo_kernel) when the problem is small enough for it: rollout,
// argmin and the winner's record without rolling the winner a second time.  ACMPC_NO_SOLO keeps the two launches.
bool use_solo(const acmpc_ctx* c, int P, int N, int n, int layout) {
  static_assert(kTraceBlocks >= acmpc::kSoloBlocks, "the trace buffer holds one trace per workgroup");
  return c->prm.mode == ACMPC_MODE_SPATIAL && !c->sw.no_solo && acmpc::solo_fits(P, N, n, layout, c->opt);
}

int solve_solo(acmpc_ctx* c, const float* d_x0, const float* d_U, int P, int N, int n, int layout, float* d_costs,
               int64_t* d_keys, float* d_records, hipStream_t s) {
  const int rc = ensure_tail_buffers(c);
  if (rc != ACMPC_OK) return rc;
  acmpc::RolloutArgs a{};
  a.U = d_U;
  a.x0 = d_x0;
  a.coef = c->d_coef;
  a.costs = d_costs;
  a.partial_keys = c->d_partial_keys;
  a.partial_feas = c->d_partial_feas;
  a.P = P;
  a.N = N;
  a.n = n;
  a.index_offset = 0;
  a.w = c->w;
  acmpc::FusedFinalize ff{};
  ff.tickets = c->d_tickets;
  ff.records = d_records;
  ff.trace = c->d_trace;
  ff.trace_pitch = acmpc::solo_trace_floats(n);
  ff.keys_out = d_keys;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof_used < c->prof_start.size()) {
    e0 = c->prof_start[c->prof_used];
    e1 = c->prof_stop[c->prof_used];
    ++c->prof_used;
  }
  ACMPC_HIP(c, acmpc::launch_rollout_solo(layout, a, ff, s, e0, e1, c->opt));
  return ACMPC_OK;
}

// The batched solve: rollout_kernel + finalize_kernel, or - ACMPC_TAILED_ROLLOUT=1, where the shape allows it (mode S,
// step-major, the 256-thread launch shapes) - both in ONE launch (rollout_tailed_kernel: the last workgroup of a problem
// finalizes it).  The same bits either way (tests/test_gpu_tailed_rollout.py).  The one launch is NOT the default: measured
// on the headline's batch (4 096 x 4 096 x 49, same box) it ends the step's launch gap - ms_per_step 1.145 against a
// kernel of 1.138 - but the kernel grows by 47 us (every workgroup's first wave waits for its ticket's round trip before
// it retires, 16 384 times, and 4 096 lone-wave re-rolls take issue slots from the streaming waves), more than the 29 us
// finalize_kernel + gap it replaces: 1.125 ms per step in two launches.  `regen`: the winner re-drawn from its index
// (counter-based candidates) instead of read from U.
int solve_batched(acmpc_ctx* c, const float* d_x0, const float* d_U, int P, int N, int n, int layout, float* d_costs,
                  int64_t* d_keys, float* d_records, hipStream_t s, const Regenerate* regen) {
  const acmpc::LaunchShape shape = acmpc::choose_shape(P, N, layout, c->prm.mode, n, c->opt);
  if (!c->sw.tailed_rollout || d_records == nullptr || !acmpc::tailed_rollout_fits(c->prm.mode, layout, shape, n)) {
    acmpc::LaunchShape used;
    int rc = rollout(c, d_x0, d_U, P, N, n, layout, 0, d_costs, s, &used);
    if (rc != ACMPC_OK) return rc;
    return finalize(c, nullptr, d_keys, d_x0, regen != nullptr ? nullptr : d_U, P, N, n, layout, 0, d_records,
                    used.blocks_per_problem, s, regen);
  }
  int rc = ensure_tail_buffers(c);
  if (rc != ACMPC_OK) return rc;
  acmpc::RolloutArgs a{};
  a.U = d_U;
  a.x0 = d_x0;
  a.coef = c->d_coef;
  a.costs = d_costs;
  a.partial_keys = c->d_partial_keys;
  a.partial_feas = c->d_partial_feas;
  a.P = P;
  a.N = N;
  a.n = n;
  a.index_offset = 0;
  a.w = c->w;
  acmpc::FinalizeArgs f{};
  if (regen != nullptr) {
    f.regenerate = true;
    f.centre = regen->d_centre;
    f.centre_stride = regen->centre_stride;
    f.u_ref = regen->d_uref;
    f.u_extra = regen->d_extra;
    f.spec = regen->spec;
  }
  f.U = d_U;
  f.x0 = d_x0;
  f.coef = c->d_coef;
  f.partial_keys = c->d_partial_keys;
  f.partial_feas = c->d_partial_feas;
  f.keys_out = d_keys;
  f.records = d_records;
  f.blocks_per_problem = shape.blocks_per_problem;
  f.P = P;
  f.N = N;
  f.n = n;
  f.index_offset = 0;
  f.w = c->w;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof_used < c->prof_start.size()) {
    e0 = c->prof_start[c->prof_used];
    e1 = c->prof_stop[c->prof_used];
    ++c->prof_used;
  }
  ACMPC_HIP(c, acmpc::launch_rollout_tailed(layout, shape, a, f, c->d_tickets, s, e0, e1));
  return ACMPC_OK;
}

int ensure_staging(acmpc_ctx* c) {
  if (c->staging_ready) return ACMPC_OK;
  c->touched_device = true;
  const acmpc_params& p = c->prm;
  const size_t cand = static_cast<size_t>(p.max_problems) * p.max_candidates;
  ACMPC_HIP(c, alloc_once(&c->d_centre, static_cast<size_t>(p.max_problems) * p.max_steps * 2 * sizeof(float)));
  ACMPC_HIP(c, alloc_once(&c->d_uref, static_cast<size_t>(p.max_problems) * p.max_steps * 2 * sizeof(float)));
  if (c->stream == nullptr) ACMPC_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  ACMPC_HIP(c, alloc_once(&c->d_U, cand * p.max_steps * 2 * sizeof(float)));
  ACMPC_HIP(c, alloc_once(&c->d_x0, static_cast<size_t>(p.max_problems) * 3 * sizeof(float)));
  ACMPC_HIP(c, alloc_once(&c->d_costs, cand * sizeof(float)));
  ACMPC_HIP(c, alloc_once(&c->d_records,
                          static_cast<size_t>(p.max_problems) * acmpc_record_floats(p.max_steps) * sizeof(float)));
  ACMPC_HIP(c, alloc_once(&c->d_keys, static_cast<size_t>(p.max_problems) * sizeof(int64_t)));
  const int rc_tail = ensure_tail_buffers(c);
  if (rc_tail != ACMPC_OK) return rc_tail;
  ACMPC_HIP(c, host_alloc_once(&c->h_keys, static_cast<size_t>(p.max_problems) * sizeof(int64_t)));
  ACMPC_HIP(c, host_alloc_once(&c->h_io, (static_cast<size_t>(p.max_problems) * (3 + acmpc_record_floats(p.max_steps)) + 4) * sizeof(float)));
  if (p.lq_candidate != 0 && c->h_lq == nullptr) {
    const size_t lq_bytes = static_cast<size_t>(p.max_problems) * p.max_steps * 2 * sizeof(float);
    ACMPC_HIP(c, host_alloc_once(&c->h_lq, lq_bytes));
    std::memset(c->h_lq, 0, lq_bytes);
  }
  c->staging_ready = true;
  return ACMPC_OK;
}

// ---- A/B switches: names as the environment spells them; a null or empty value, or "0" for the boolean ones, is the default
const char* const kOptionNames[] = {
    "ACMPC_SHAPE", "ACMPC_T_PACK", "ACMPC_NO_TILE", "ACMPC_TILE_ROWS", "ACMPC_TILE_TABLE", "ACMPC_NO_TRIO_ROUNDS",
    "ACMPC_NO_QUAD_ROUNDS", "ACMPC_NO_PAIR_ROUNDS", "ACMPC_SOLO_REGISTERS", "ACMPC_SOLO_SPLIT", "ACMPC_NO_VERIFIED_SEARCH",
    "ACMPC_NO_SOLO", "ACMPC_NO_FUSED_FINALIZE", "ACMPC_NO_TRACED_FINALIZE", "ACMPC_NO_CHAINED_ROUNDS", "ACMPC_NO_GRAPH",
    "ACMPC_NO_FUSED_SAMPLING", "ACMPC_TICK_GRAPH", "ACMPC_TICK_NO_FLAG", "ACMPC_TICK_NO_INLINE_PATH", "ACMPC_NO_ZERO_COPY", "ACMPC_TAILED_ROLLOUT", "ACMPC_NO_GROUP_FINALIZE", "ACMPC_FINALIZE_WAVES",
    "ACMPC_NO_CHAINED_STREAM", "ACMPC_LQ_BOX_ITERATIONS", "ACMPC_START_CLOCKS",
    "ACMPC_CONFORMANT_SYNC"};   // (last: it sets several of the switches above, and wins over them when both are in the environment)

bool apply_option(acmpc_ctx* c, const char* name, const char* value) {
  const std::string key(name);
  const bool present = value != nullptr && value[0] != '\0';
  const bool on = present && !(value[0] == '0' && value[1] == '\0');
  auto tri = [&](int* field) { *field = present ? (value[0] == '1' ? 1 : 0) : -1; return true; };
  acmpc::LaunchOptions& o = c->opt;
  if (key == "ACMPC_SHAPE") {
    o.shape_block = o.shape_cpt = 0;
    if (present && std::sscanf(value, "%d,%d", &o.shape_block, &o.shape_cpt) != 2) o.shape_block = o.shape_cpt = 0;
    return true;
  }
  if (key == "ACMPC_T_PACK") { o.temporal_pack = present ? (value[0] == '1' ? 1 : 2) : 0; return true; }
  if (key == "ACMPC_NO_TILE") { o.no_tile = on; return true; }
  if (key == "ACMPC_TILE_ROWS") { o.tile_rows = present ? std::atoi(value) : -1; return true; }
  if (key == "ACMPC_TILE_TABLE") { o.tile_table = present ? (value[0] == 'l' ? 1 : 2) : 0; return true; }
  if (key == "ACMPC_NO_TRIO_ROUNDS") { o.no_trio_rounds = on; return true; }
  if (key == "ACMPC_NO_QUAD_ROUNDS") { o.no_quad_rounds = on; return true; }
  if (key == "ACMPC_NO_PAIR_ROUNDS") { o.no_pair_rounds = on; return true; }
  if (key == "ACMPC_SOLO_REGISTERS") return tri(&o.solo_registers);
  if (key == "ACMPC_SOLO_SPLIT") return tri(&o.solo_split);
  if (key == "ACMPC_NO_GROUP_FINALIZE") { o.no_group_finalize = on; return true; }
  if (key == "ACMPC_FINALIZE_WAVES") { o.finalize_waves = on; return true; }
  acmpc_ctx::Switches& w = c->sw;
  if (key == "ACMPC_NO_VERIFIED_SEARCH") { w.no_verified_search = on; return true; }
  if (key == "ACMPC_NO_SOLO") { w.no_solo = on; return true; }
  if (key == "ACMPC_NO_FUSED_FINALIZE") { w.no_fused_finalize = on; return true; }
  if (key == "ACMPC_NO_TRACED_FINALIZE") { w.no_traced_finalize = on; return true; }
  if (key == "ACMPC_NO_CHAINED_ROUNDS") { w.no_chained_rounds = on; return true; }
  if (key == "ACMPC_NO_CHAINED_STREAM") { w.no_chained_stream = on; return true; }
  if (key == "ACMPC_NO_GRAPH") { w.no_graph = on; return true; }
  if (key == "ACMPC_NO_FUSED_SAMPLING") { w.no_fused_sampling = on; return true; }
  if (key == "ACMPC_TICK_GRAPH") { w.tick_graph = on; return true; }
  if (key == "ACMPC_TICK_NO_FLAG") { w.tick_no_flag = on; return true; }
  if (key == "ACMPC_TICK_NO_INLINE_PATH") { w.tick_no_inline_path = on; return true; }
  if (key == "ACMPC_NO_ZERO_COPY") { w.no_zero_copy = on; return true; }
  if (key == "ACMPC_TAILED_ROLLOUT") { w.tailed_rollout = on; return true; }
  if (key == "ACMPC_START_CLOCKS") { c->want_start_clocks = on; return true; }
  if (key == "ACMPC_CONFORMANT_SYNC") {
    // ONE switch for the forms that stay inside the HSA memory model and HIP's barrier rule (include/acmpc.h): every solve,
    // round and batch as separate launches, nothing published between workgroups of one launch, no wave of a workgroup
    // ending while the others still meet at a barrier, completion by hipStreamSynchronize.  Sets (or, off, clears) the
    // switches that select them - the one-launch solve, the in-launch finalize (with it the traced finalize, the chained
    // rounds and the multi-wave rounds, which need it), the chained stream, the tick's completion flag - and the tailed
    // rollout off.  The same bits either way (tests/test_gpu_conformant.py); INTEGRATION.md section 6 has what it costs.
    w.no_solo = w.no_fused_finalize = w.no_chained_stream = w.tick_no_flag = on;
    if (on) w.tailed_rollout = false;
    return true;
  }
  if (key == "ACMPC_LQ_BOX_ITERATIONS") { c->lq_box_iterations = present ? std::max(0, std::atoi(value)) : 40; return true; }
  return false;
}

}  // namespace

extern "C" {

const char* acmpc_version(void) { return "acmpc-hip 0.1 gfx950"; }

int32_t acmpc_record_floats(int32_t n) { return ACMPC_REC_HEADER + 2 * n + 3 * (n + 1); }

int acmpc_lq_plan(const double* table, int32_t n, const double x0[3], const double step_cost[3], const double r_term[2],
                  const double final_cost[3], const float u_min[2], const float u_max[2], float* plan) {
  if (table == nullptr || x0 == nullptr || step_cost == nullptr || r_term == nullptr || final_cost == nullptr ||
      u_min == nullptr || u_max == nullptr || plan == nullptr || n < 1)
    return ACMPC_EINVAL;
  return acmpc::lq::plan(table, n, x0, step_cost, r_term, final_cost, u_min, u_max, plan) ? ACMPC_OK : ACMPC_ESTATE;
}

int acmpc_lq_box_plan(const double* table, int32_t n, const double x0[3], const double step_cost[3], const double r_term[2],
                      const double final_cost[3], const float u_min[2], const float u_max[2], double margin, double w_bound,
                      int32_t iterations, double* state, float* plan, double* info) {
  if (table == nullptr || x0 == nullptr || step_cost == nullptr || r_term == nullptr || final_cost == nullptr ||
      u_min == nullptr || u_max == nullptr || plan == nullptr || state == nullptr || info == nullptr || n < 1)
    return ACMPC_EINVAL;
  if (!acmpc::lq::plan(table, n, x0, step_cost, r_term, final_cost, u_min, u_max, plan)) return ACMPC_ESTATE;
  acmpc::lqbox::State st;
  const size_t m = static_cast<size_t>(n) * 2;
  if (state[0] == static_cast<double>(n)) {   // a warm iterate: wx, wu, lx, lu behind the horizon it belongs to
    st.n = n;
    st.wx.assign(state + 1, state + 1 + m);
    st.wu.assign(state + 1 + m, state + 1 + 2 * m);
    st.lx.assign(state + 1 + 2 * m, state + 1 + 3 * m);
    st.lu.assign(state + 1 + 3 * m, state + 1 + 4 * m);
  }
  acmpc::lqbox::Workspace ws;
  const acmpc::lqbox::Result r = acmpc::lqbox::refine(table, n, x0, step_cost, r_term, final_cost, u_min, u_max, margin,
                                                      w_bound, iterations, st, ws, plan);
  state[0] = static_cast<double>(st.n);
  if (st.n == n) {
    std::copy(st.wx.begin(), st.wx.end(), state + 1);
    std::copy(st.wu.begin(), st.wu.end(), state + 1 + m);
    std::copy(st.lx.begin(), st.lx.end(), state + 1 + 2 * m);
    std::copy(st.lu.begin(), st.lu.end(), state + 1 + 3 * m);
  }
  info[0] = r.iterations, info[1] = r.chosen, info[2] = r.triggered ? 1.0 : 0.0, info[3] = r.cost.J, info[4] = r.cost.V;
  return ACMPC_OK;
}

int acmpc_lq_box_stats(const acmpc_ctx* c, double info[5]) {
  if (c == nullptr || info == nullptr) return ACMPC_EINVAL;
  const acmpc::lqbox::Result& r = c->lq_box_last;
  info[0] = r.iterations, info[1] = r.chosen, info[2] = r.triggered ? 1.0 : 0.0, info[3] = r.cost.J, info[4] = r.cost.V;
  return ACMPC_OK;
}

int64_t acmpc_pack_key(float cost, uint32_t index) { return acmpc::pack_key(cost, index); }

float acmpc_key_cost(int64_t key) {
  const int32_t hi = static_cast<int32_t>(key >> 32);
  const int32_t bits = (hi >= 0) ? hi : (hi ^ 0x7fffffff);
  float f;
  std::memcpy(&f, &bits, sizeof f);
  return f;
}

uint32_t acmpc_key_index(int64_t key) { return static_cast<uint32_t>(key & 0xffffffffLL); }

const char* acmpc_last_error(const acmpc_ctx* ctx) { return ctx != nullptr ? ctx->err.c_str() : g_create_error.c_str(); }

int acmpc_create(const acmpc_params* params, acmpc_ctx** out) {
  if (params == nullptr || out == nullptr) return fail(nullptr, ACMPC_EINVAL, "null argument");
  *out = nullptr;
  if (params->struct_size != sizeof(acmpc_params)) return fail(nullptr, ACMPC_EINVAL, "acmpc_params size mismatch");
  if (params->mode != ACMPC_MODE_SPATIAL && params->mode != ACMPC_MODE_TEMPORAL)
    return fail(nullptr, ACMPC_EINVAL, "unknown mode");
  if (params->max_problems < 1 || params->max_candidates < 1 || params->max_steps < 1)
    return fail(nullptr, ACMPC_EINVAL, "capacities must be positive");
  if (params->nn_ahead >= 0 && (params->nn_back < 0 || params->nn_back + params->nn_ahead + 1 > 64))
    return fail(nullptr, ACMPC_EINVAL, "nearest-waypoint window: need nn_back >= 0 and at most 64 waypoints");
  if (params->centre_update != 0 && params->centre_update != 1)
    return fail(nullptr, ACMPC_EINVAL, "centre_update must be 0 (argmin) or 1 (softmin mean)");
  if (params->lq_candidate < 0 || params->lq_candidate > 2)
    return fail(nullptr, ACMPC_EINVAL, "lq_candidate must be 0, 1 (LQ plan) or 2 (LQ plan + box-constrained refinement)");
  if (params->max_steps > 1024)
    return fail(nullptr, ACMPC_EINVAL, "the waypoint table and the winner record are staged in LDS: max_steps <= 1024");
  if (params->max_problems > 65535)
    return fail(nullptr, ACMPC_EINVAL, "problems map to the grid's y dimension: max_problems <= 65535");
  acmpc_ctx* c = new (std::nothrow) acmpc_ctx();
  if (c == nullptr) return fail(nullptr, ACMPC_EINVAL, "out of host memory");
  c->prm = *params;
  c->coef_stride = params->mode == ACMPC_MODE_SPATIAL ? ACMPC_COEF_STRIDE_SPATIAL : ACMPC_COEF_STRIDE_TEMPORAL;
  acmpc::Weights& w = c->w;
  w.q0 = static_cast<float>(params->step_cost[0]);
  w.q1 = static_cast<float>(params->step_cost[1]);
  w.q2 = static_cast<float>(params->step_cost[2]);
  w.r0 = static_cast<float>(params->r_term[0]);
  w.r1 = static_cast<float>(params->r_term[1]);
  w.qn0 = static_cast<float>(params->final_cost[0]);
  w.qn1 = static_cast<float>(params->final_cost[1]);
  w.qn2 = static_cast<float>(params->final_cost[2]);
  w.hq0 = 0.5f * w.q0;
  w.hq1 = 0.5f * w.q1;
  w.hr0 = 0.5f * w.r0;
  w.hr1 = 0.5f * w.r1;
  w.hqn0 = 0.5f * w.qn0;
  w.hqn1 = 0.5f * w.qn1;
  w.hqn2 = 0.5f * w.qn2;
  w.ulo0 = static_cast<float>(params->u_min[0]);
  w.ulo1 = static_cast<float>(params->u_min[1]);
  w.uhi0 = static_cast<float>(params->u_max[0]);
  w.uhi1 = static_cast<float>(params->u_max[1]);
  w.tmin = static_cast<float>(params->t_min);
  w.wbound = static_cast<float>(params->w_bound);
  w.dt = static_cast<float>(params->dt);
  w.nn_back = params->nn_back;
  w.nn_ahead = params->nn_ahead;
  // the A/B switches of the tests and the tools (tools/README.md): the environment is read HERE, once per handle
  for (const char* name : kOptionNames) {
    const char* value = std::getenv(name);
    if (value != nullptr) (void)apply_option(c, name, value);
  }
  *out = c;
  return ACMPC_OK;
}

int acmpc_set_option(acmpc_ctx* c, const char* name, const char* value) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (name == nullptr) return fail(c, ACMPC_EINVAL, "null option name");
  if (!apply_option(c, name, value)) return fail(c, ACMPC_EINVAL, std::string("unknown option or bad value: ") + name);
  // captured graphs hold the launch forms they were captured with
  for (hipGraphExec_t& g : c->opt_graph) {
    if (g != nullptr) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  for (hipGraphExec_t& g : c->tick_graph) {
    if (g != nullptr) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  return ACMPC_OK;
}

void acmpc_destroy(acmpc_ctx* c) {
  if (c == nullptr) return;
  if (c->touched_device) {  // also after a failed bring-up: whatever was allocated before the failure is freed
    if (c->prm.device >= 0) (void)hipSetDevice(c->prm.device);
    (void)hipFree(c->d_coef);
    (void)hipFree(c->d_start_clock);
    (void)hipFree(c->d_partial_keys);
    (void)hipFree(c->d_partial_feas);
    (void)hipFree(c->d_soft_partial);
    (void)hipFree(c->d_segments);
    (void)hipFree(c->d_centre);
    (void)hipFree(c->d_uref);
    (void)hipFree(c->d_U);
    (void)hipFree(c->d_x0);
    (void)hipFree(c->d_costs);
    (void)hipFree(c->d_records);
    (void)hipFree(c->d_keys);
    (void)hipFree(c->d_tickets);
    (void)hipFree(c->d_trace);
    (void)hipFree(c->d_nn_frames);
    if (c->h_keys != nullptr) (void)hipHostFree(c->h_keys);
    if (c->h_io != nullptr) (void)hipHostFree(c->h_io);
    if (c->h_lq != nullptr) (void)hipHostFree(c->h_lq);
    for (hipGraphExec_t g : c->opt_graph)
      if (g != nullptr) (void)hipGraphExecDestroy(g);
    for (hipGraphExec_t g : c->tick_graph)
      if (g != nullptr) (void)hipGraphExecDestroy(g);
    if (c->h_tick != nullptr) (void)hipHostFree(c->h_tick);
    if (c->h_tick_out != nullptr) (void)hipHostFree(c->h_tick_out);
    (void)hipFree(c->d_tick);
    (void)hipFree(c->d_warm);
    (void)hipFree(c->d_map);
    (void)hipFree(c->d_coords);
    if (c->h_opt != nullptr) (void)hipHostFree(c->h_opt);
    if (c->h_opt_records != nullptr) (void)hipHostFree(c->h_opt_records);
    (void)hipFree(c->d_seed);
    (void)hipFree(c->d_opt);
    if (c->stream != nullptr) (void)hipStreamDestroy(c->stream);
    for (hipEvent_t e : c->prof_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->prof_stop) (void)hipEventDestroy(e);
  }
  delete c;
}

int acmpc_set_paths(acmpc_ctx* c, const double* tables, int32_t P, int32_t n) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (tables == nullptr) return fail(c, ACMPC_EINVAL, "null tables");
  if (P < 1 || n < 2) return fail(c, ACMPC_EINVAL, "need P >= 1 and n >= 2");
  if (P > c->prm.max_problems || n > c->prm.max_steps) return fail(c, ACMPC_ECAPACITY, "P or n exceeds capacity");
  const int stride = c->coef_stride;
  c->h_coef.assign(static_cast<size_t>(P) * n * stride, 0.0f);
  const double margin = c->prm.margin;
  for (int p = 0; p < P; ++p) {
    const double* t = tables + static_cast<size_t>(p) * 7 * n;
    const double *x = t, *y = t + n, *psi = t + 2 * n, *kappa = t + 3 * n, *ds = t + 4 * n, *width = t + 5 * n,
                 *v = t + 6 * n;
    float* out = c->h_coef.data() + static_cast<size_t>(p) * n * stride;
    for (int i = 0; i < n; ++i, out += stride) {
      if (c->prm.mode == ACMPC_MODE_SPATIAL) {
        // non-trivial entries of A_i, B_i, f_i (dynamics.py:65-103) and the corridor of x_{i+1} (control.py:57-60)
        const double vds = v[i] * ds[i] + kEps;
        out[0] = static_cast<float>(ds[i]);
        out[1] = static_cast<float>(-(kappa[i] * kappa[i]) * ds[i]);
        out[2] = static_cast<float>(-kappa[i] / vds);
        out[3] = static_cast<float>(-1.0 / (v[i] * v[i] * ds[i] + kEps));
        out[4] = static_cast<float>(1.0 / vds);
        out[5] = static_cast<float>(v[i]);
        out[6] = static_cast<float>(kappa[i]);
        out[7] = static_cast<float>(-width[i] / 2.0 + margin);
        out[8] = static_cast<float>(width[i] / 2.0 - margin);
      } else {
        out[0] = static_cast<float>(x[i]);
        out[1] = static_cast<float>(y[i]);
        out[2] = static_cast<float>(std::cos(psi[i]));
        out[3] = static_cast<float>(std::sin(psi[i]));
        out[4] = static_cast<float>(psi[i]);
        out[5] = static_cast<float>(kappa[i]);
        out[6] = static_cast<float>(v[i]);
        out[7] = static_cast<float>(width[i] / 2.0 - margin);
      }
    }
  }
  if (c->prm.lq_candidate != 0) c->h_tables.assign(tables, tables + static_cast<size_t>(P) * 7 * n);
  c->h_nn_frames.clear();
  if (c->prm.mode == ACMPC_MODE_TEMPORAL && c->prm.nn_ahead < 0 && n >= acmpc::kVerifiedWindow && n <= kMaxVerifiedSteps)
    verified_frames(c->h_coef.data(), P, n, &c->h_nn_frames);
  c->P_set = P;
  c->n_set = n;
  c->tables_dirty = true;
  c->frames_dirty = !c->h_nn_frames.empty();
  return ACMPC_OK;
}

int acmpc_set_coefficients(acmpc_ctx* c, const float* coef, int32_t P, int32_t n) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (coef == nullptr) return fail(c, ACMPC_EINVAL, "null coefficients");
  if (P < 1 || n < 2) return fail(c, ACMPC_EINVAL, "need P >= 1 and n >= 2");
  if (P > c->prm.max_problems || n > c->prm.max_steps) return fail(c, ACMPC_ECAPACITY, "P or n exceeds capacity");
  c->h_coef.assign(coef, coef + static_cast<size_t>(P) * n * c->coef_stride);
  if (c->h_tables.size() != static_cast<size_t>(P) * 7 * n) c->h_tables.clear();   // (no float64 tables for these paths)
  c->h_nn_frames.clear();
  if (c->prm.mode == ACMPC_MODE_TEMPORAL && c->prm.nn_ahead < 0 && n >= acmpc::kVerifiedWindow && n <= kMaxVerifiedSteps)
    verified_frames(c->h_coef.data(), P, n, &c->h_nn_frames);
  c->P_set = P;
  c->n_set = n;
  c->tables_dirty = true;
  c->frames_dirty = !c->h_nn_frames.empty();
  return ACMPC_OK;
}

int32_t acmpc_search_window(int32_t* back) {
  if (back != nullptr) *back = acmpc::kVerifiedBack;
  return acmpc::kVerifiedWindow;
}

int32_t acmpc_search_frame_floats(int32_t n) {
  return n >= acmpc::kVerifiedWindow ? acmpc::verified_frame_floats(n) : 0;
}

int acmpc_search_frames(const float* coef, int32_t P, int32_t n, float* out, int64_t capacity_floats) {
  if (coef == nullptr || out == nullptr || P < 1 || n < acmpc::kVerifiedWindow) return ACMPC_EINVAL;
  std::vector<float> frames;
  verified_frames(coef, P, n, &frames);
  if (capacity_floats < static_cast<int64_t>(frames.size())) return ACMPC_ECAPACITY;
  std::memcpy(out, frames.data(), frames.size() * sizeof(float));
  return ACMPC_OK;
}

int acmpc_get_coefficients(const acmpc_ctx* c, int32_t problem, float* out, int32_t capacity_floats) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (out == nullptr) return fail(c, ACMPC_EINVAL, "null output");
  if (c->P_set == 0) return fail(c, ACMPC_ESTATE, "acmpc_set_paths has not been called");
  if (problem < 0 || problem >= c->P_set) return fail(c, ACMPC_EINVAL, "problem index out of range");
  const size_t count = static_cast<size_t>(c->n_set) * c->coef_stride;
  if (capacity_floats < static_cast<int64_t>(count)) return fail(c, ACMPC_ECAPACITY, "output buffer too small");
  std::memcpy(out, c->h_coef.data() + static_cast<size_t>(problem) * count, count * sizeof(float));
  return ACMPC_OK;
}

int acmpc_sync_tables(acmpc_ctx* c, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (c->stream_pending && (c->tables_dirty || c->frames_dirty))   // (the pending finalize reads the tables on the device)
    return fail(c, ACMPC_ESTATE, "a batch of acmpc_solve_stream_device is pending: acmpc_solve_stream_flush first");
  int rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = upload_tables(c, s);
  if (rc != ACMPC_OK) return rc;
  ACMPC_HIP(c, hipStreamSynchronize(s));
  return ACMPC_OK;
}

int acmpc_rollout_device(acmpc_ctx* c, const float* d_x0, const float* d_U, int32_t P, int32_t N, int32_t n,
                         int32_t layout, int64_t index_offset, float* d_costs, int64_t* d_keys, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_x0 == nullptr || d_U == nullptr) return fail(c, ACMPC_EINVAL, "null device pointer");
  if (index_offset < 0 || index_offset + N > 0xffffffffLL) return fail(c, ACMPC_EINVAL, "global index exceeds 32 bits");
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = upload_tables(c, s);
  if (rc != ACMPC_OK) return rc;
  acmpc::LaunchShape shape;
  rc = rollout(c, d_x0, d_U, P, N, n, layout, index_offset, d_costs, s, &shape);
  if (rc != ACMPC_OK || d_keys == nullptr) return rc;
  return finalize(c, nullptr, d_keys, d_x0, d_U, P, N, n, layout, index_offset, nullptr, shape.blocks_per_problem, s);
}

int acmpc_finalize_device(acmpc_ctx* c, const int64_t* d_keys, const float* d_x0, const float* d_U, int32_t P,
                          int32_t N, int32_t n, int32_t layout, int64_t index_offset, float* d_records,
                          void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_x0 == nullptr || d_U == nullptr || d_records == nullptr) return fail(c, ACMPC_EINVAL, "null device pointer");
  if (index_offset < 0 || index_offset + N > 0xffffffffLL) return fail(c, ACMPC_EINVAL, "global index exceeds 32 bits");
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  if (!c->device_ready) return fail(c, ACMPC_ESTATE, "acmpc_rollout_device must run first");
  const acmpc::LaunchShape shape = acmpc::choose_shape(P, N, layout, c->prm.mode, n, c->opt);
  return finalize(c, d_keys, nullptr, d_x0, d_U, P, N, n, layout, index_offset, d_records, shape.blocks_per_problem,
                  static_cast<hipStream_t>(stream));
}

int acmpc_solve_device(acmpc_ctx* c, const float* d_x0, const float* d_U, int32_t P, int32_t N, int32_t n,
                       int32_t layout, float* d_costs, int64_t* d_keys, float* d_records, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_x0 == nullptr || d_U == nullptr) return fail(c, ACMPC_EINVAL, "null device pointer");
  if (d_keys == nullptr && d_records == nullptr) return fail(c, ACMPC_EINVAL, "need d_keys and/or d_records");
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = upload_tables(c, s);
  if (rc != ACMPC_OK) return rc;
  if (use_solo(c, P, N, n, layout)) return solve_solo(c, d_x0, d_U, P, N, n, layout, d_costs, d_keys, d_records, s);
  return solve_batched(c, d_x0, d_U, P, N, n, layout, d_costs, d_keys, d_records, s, nullptr);
}

int acmpc_solve_sampled_device(acmpc_ctx* c, const float* d_x0, const float* d_U, const float* d_centre,
                               int32_t centre_stride, const float* d_u_ref, int32_t P, int32_t N, int32_t n, int32_t layout,
                               double sigma_v, double sigma_kappa, uint64_t seed, uint32_t round, float* d_costs,
                               int64_t* d_keys, float* d_records, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_x0 == nullptr || d_U == nullptr || d_centre == nullptr || d_records == nullptr)
    return fail(c, ACMPC_EINVAL, "null device pointer");
  if (centre_stride < 2 * n) return fail(c, ACMPC_EINVAL, "centre_stride must be at least 2 n");
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = upload_tables(c, s);
  if (rc != ACMPC_OK) return rc;
  rc = upload_segments(c, n, s);
  if (rc != ACMPC_OK) return rc;
  const Regenerate regen{d_centre, centre_stride, d_u_ref, make_spec(c, sigma_v, sigma_kappa, seed, round)};
  return solve_batched(c, d_x0, d_U, P, N, n, layout, d_costs, d_keys, d_records, s, &regen);
}

int acmpc_solve_stream_flush(acmpc_ctx* c, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (!c->stream_pending) return ACMPC_OK;
  c->stream_pending = false;
  ACMPC_HIP(c, acmpc::launch_finalize(c->prm.mode, c->stream_fin_layout, c->stream_fin, static_cast<hipStream_t>(stream), c->opt));
  return ACMPC_OK;
}

int acmpc_solve_stream_device(acmpc_ctx* c, const float* d_x0, const float* d_U, const float* d_centre,
                              int32_t centre_stride, const float* d_u_ref, int32_t P, int32_t N, int32_t n, int32_t layout,
                              double sigma_v, double sigma_kappa, uint64_t seed, uint32_t round, float* d_costs,
                              int64_t* d_keys, float* d_records, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_x0 == nullptr || d_U == nullptr || d_records == nullptr) return fail(c, ACMPC_EINVAL, "null device pointer");
  if (d_centre != nullptr && centre_stride < 2 * n) return fail(c, ACMPC_EINVAL, "centre_stride must be at least 2 n");
  int rc = check_shape(c, P, N, n, layout, true);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the pending finalize reads the tables its batch was rolled with: new ones go up behind it
  // (and the sampler's knot table of its horizon: upload_segments rewrites it in place for another)
  if (c->stream_pending && (c->tables_dirty || c->frames_dirty || (d_centre != nullptr && c->segments_n != n))) {
    rc = acmpc_solve_stream_flush(c, stream);
    if (rc != ACMPC_OK) return rc;
  }
  rc = upload_tables(c, s);
  if (rc != ACMPC_OK) return rc;
  if (d_centre != nullptr) {
    rc = upload_segments(c, n, s);
    if (rc != ACMPC_OK) return rc;
  }
  const acmpc::LaunchShape shape = acmpc::choose_shape(P, N, layout, c->prm.mode, n, c->opt);
  const int set = c->stream_pending ? (c->stream_set ^ 1) : 0;
  acmpc::RolloutArgs a{};
  a.U = d_U;
  a.x0 = d_x0;
  a.coef = c->d_coef;
  a.nn_frames = (!c->h_nn_frames.empty() && !c->sw.no_verified_search) ? c->d_nn_frames : nullptr;
  a.costs = d_costs;
  a.partial_keys = c->d_partial_keys + set * c->partial_slots;
  a.partial_feas = c->d_partial_feas + set * c->partial_slots;
  a.P = P;
  a.N = N;
  a.n = n;
  a.index_offset = 0;
  a.w = c->w;
  acmpc::FinalizeArgs f{};
  if (d_centre != nullptr) {
    f.regenerate = true;
    f.centre = d_centre;
    f.centre_stride = centre_stride;
    f.u_ref = d_u_ref;
    f.spec = make_spec(c, sigma_v, sigma_kappa, seed, round);
  }
  f.U = d_centre != nullptr ? nullptr : d_U;
  f.x0 = d_x0;
  f.coef = c->d_coef;
  f.partial_keys = a.partial_keys;
  f.partial_feas = a.partial_feas;
  f.keys_out = d_keys;
  f.records = d_records;
  f.blocks_per_problem = shape.blocks_per_problem;
  f.P = P;
  f.N = N;
  f.n = n;
  f.index_offset = 0;
  f.w = c->w;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->prof_used < c->prof_start.size()) {
    e0 = c->prof_start[c->prof_used];
    e1 = c->prof_stop[c->prof_used];
    ++c->prof_used;
  }
  if (c->stream_pending &&
      acmpc::chained_rollout_fits(c->prm.mode, layout, shape, P, c->stream_fin, c->stream_fin_layout) && !c->sw.no_chained_stream) {
    c->stream_pending = false;
    ACMPC_HIP(c, acmpc::launch_rollout_chained(layout, shape, a, c->stream_fin, c->stream_fin_layout, s, e0, e1));
  } else {
    rc = acmpc_solve_stream_flush(c, stream);
    if (rc != ACMPC_OK) return rc;
    ACMPC_HIP(c, acmpc::launch_rollout(c->prm.mode, layout, shape, a, s, e0, e1));
  }
  c->stream_fin = f;
  c->stream_fin_layout = layout;
  c->stream_set = set;
  c->stream_pending = true;
  return ACMPC_OK;
}

int acmpc_solve(acmpc_ctx* c, const float* x0, const float* U, int32_t P, int32_t N, int32_t n, int32_t layout,
                float* costs, int32_t* best_idx, float* records) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (x0 == nullptr || U == nullptr) return fail(c, ACMPC_EINVAL, "null input");
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_staging(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = c->stream;
  rc = upload_tables(c, s);
  if (rc != ACMPC_OK) return rc;
  const size_t cand = static_cast<size_t>(P) * N;
  const size_t rec_bytes = static_cast<size_t>(P) * acmpc_record_floats(n) * sizeof(float);
  // Nothing small crosses the host link as a copy of its own (round 4): the start states are written into the handle's
  // page-locked block and READ THERE by the kernels, keys and records are written there BY the kernels (page-locked host
  // memory is device-addressable: what acmpc_control_tick does with its tick block) - each of those copies was a packet of
  // ~4 us in the stream.  The control matrix is read in place too when the caller built it in page-locked memory
  // (acmpc_host_alloc): the rollout then streams it over the host link while it computes, instead of behind a copy of the
  // whole matrix; from pageable memory it is staged into device memory as before.  ACMPC_NO_ZERO_COPY=1: every transfer a copy.
  float* h_x0 = c->h_io;
  float* h_records = c->h_io + ((static_cast<size_t>(P) * 3 + 3) & ~static_cast<size_t>(3));   // (16-byte aligned)
  std::memcpy(h_x0, x0, static_cast<size_t>(P) * 3 * sizeof(float));
  const bool solo = use_solo(c, P, N, n, layout);
  // (the one-launch solve: a few workgroups, latency is everything.  A batch of thousands of problems keeps its small
  // copies - every workgroup fetching its start state over the host link would be thousands of requests for one packet)
  const bool in_place = !c->sw.no_zero_copy && solo;
  const float* d_x0 = c->d_x0;
  const float* d_U = c->d_U;
  int64_t* d_keys = c->d_keys;
  float* d_records = records != nullptr ? c->d_records : nullptr;
  if (in_place) {
    d_x0 = h_x0;
    d_keys = c->h_keys;
    if (records != nullptr) d_records = h_records;
  } else {
    ACMPC_HIP(c, hipMemcpyAsync(c->d_x0, h_x0, static_cast<size_t>(P) * 3 * sizeof(float), hipMemcpyHostToDevice, s));
  }
  if (!c->sw.no_zero_copy) {
    hipPointerAttribute_t where{};
    if (hipPointerGetAttributes(&where, U) == hipSuccess && where.type == hipMemoryTypeHost && where.devicePointer != nullptr) {
      d_U = static_cast<const float*>(where.devicePointer);
    } else {
      (void)hipGetLastError();   // (pageable memory is not an error here)
    }
  }
  if (d_U == c->d_U) ACMPC_HIP(c, hipMemcpyAsync(c->d_U, U, cand * n * 2 * sizeof(float), hipMemcpyHostToDevice, s));
  if (solo) {
    rc = solve_solo(c, d_x0, d_U, P, N, n, layout, costs != nullptr ? c->d_costs : nullptr, d_keys, d_records, s);
  } else {
    acmpc::LaunchShape shape;
    rc = rollout(c, d_x0, d_U, P, N, n, layout, 0, costs != nullptr ? c->d_costs : nullptr, s, &shape);
    if (rc != ACMPC_OK) return rc;
    rc = finalize(c, nullptr, d_keys, d_x0, d_U, P, N, n, layout, 0, d_records, shape.blocks_per_problem, s);
  }
  if (rc != ACMPC_OK) return rc;
  if (!in_place) {
    ACMPC_HIP(c, hipMemcpyAsync(c->h_keys, c->d_keys, static_cast<size_t>(P) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (records != nullptr) ACMPC_HIP(c, hipMemcpyAsync(h_records, c->d_records, rec_bytes, hipMemcpyDeviceToHost, s));
  }
  if (costs != nullptr) ACMPC_HIP(c, hipMemcpyAsync(costs, c->d_costs, cand * sizeof(float), hipMemcpyDeviceToHost, s));
  ACMPC_HIP(c, hipStreamSynchronize(s));
  if (records != nullptr) std::memcpy(records, h_records, rec_bytes);
  if (best_idx != nullptr)
    for (int p = 0; p < P; ++p) best_idx[p] = static_cast<int32_t>(acmpc_key_index(c->h_keys[p]));
  return ACMPC_OK;
}

int acmpc_host_alloc(void** out, uint64_t bytes) {
  if (out == nullptr || bytes == 0) return fail(nullptr, ACMPC_EINVAL, "acmpc_host_alloc: null output or zero size");
  *out = nullptr;
  const hipError_t e = hipHostMalloc(out, static_cast<size_t>(bytes), hipHostMallocDefault);
  if (e != hipSuccess) return fail_hip(nullptr, e, "hipHostMalloc");
  return ACMPC_OK;
}

int acmpc_host_free(void* memory) {
  if (memory == nullptr) return ACMPC_OK;
  const hipError_t e = hipHostFree(memory);
  if (e != hipSuccess) return fail_hip(nullptr, e, "hipHostFree");
  return ACMPC_OK;
}

void acmpc_philox4x32(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  acmpc::philox4x32_10(counter, key, out);
}

int acmpc_sample_device(acmpc_ctx* c, const float* d_centre, int32_t centre_stride, const float* d_u_ref, int32_t P,
                        int32_t N, int32_t n, int32_t layout, int64_t index_offset, double sigma_v, double sigma_kappa,
                        uint64_t seed, uint32_t round, float* d_U, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_centre == nullptr || d_U == nullptr) return fail(c, ACMPC_EINVAL, "null device pointer");
  if (centre_stride < 2 * n) return fail(c, ACMPC_EINVAL, "centre_stride must be at least 2 n");
  if (index_offset < 0 || index_offset + N > 0xffffffffLL) return fail(c, ACMPC_EINVAL, "global index exceeds 32 bits");
  // (allowed while a batch of acmpc_solve_stream_device is pending: drawing the next batch's candidates touches neither
  // the tables nor the partial keys the pending finalize reads)
  int rc = check_shape(c, P, N, n, layout, true);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  return sample(c, d_centre, centre_stride, d_u_ref, P, N, n, layout, index_offset, sigma_v, sigma_kappa, seed, round,
                d_U, static_cast<hipStream_t>(stream));
}

int acmpc_finalize_sampled_device(acmpc_ctx* c, const int64_t* d_keys, const float* d_x0, const float* d_centre,
                                  int32_t centre_stride, const float* d_u_ref, int32_t P, int32_t N, int32_t n,
                                  double sigma_v, double sigma_kappa, uint64_t seed, uint32_t round, float* d_records,
                                  void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_x0 == nullptr || d_centre == nullptr || d_records == nullptr) return fail(c, ACMPC_EINVAL, "null device pointer");
  if (centre_stride < 2 * n) return fail(c, ACMPC_EINVAL, "centre_stride must be at least 2 n");
  int rc = check_shape(c, P, N, n, ACMPC_LAYOUT_STEP_MAJOR);
  if (rc != ACMPC_OK) return rc;
  if (!c->device_ready) return fail(c, ACMPC_ESTATE, "acmpc_rollout_device must run first");
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = upload_segments(c, n, s);
  if (rc != ACMPC_OK) return rc;
  Regenerate regen{d_centre, centre_stride, d_u_ref, make_spec(c, sigma_v, sigma_kappa, seed, round)};
  const acmpc::LaunchShape shape = acmpc::choose_shape(P, N, ACMPC_LAYOUT_STEP_MAJOR, c->prm.mode, n, c->opt);
  return finalize(c, d_keys, nullptr, d_x0, nullptr, P, N, n, ACMPC_LAYOUT_STEP_MAJOR, 0, d_records,
                  shape.blocks_per_problem, s, &regen);
}

extern "C++" {
namespace {

// the launch sequence of one optimisation, enqueued on `s` (directly, or while `s` is being captured): per round ONE
// fused sample + rollout launch (candidates never touch memory) and the finalize that re-draws the winner from its
// index; `fused == false` keeps the three-kernel form (sample -> U -> rollout -> finalize), which the tests compare
struct OptInputs {
  const float* x0;
  const float* centre;
  const float* uref;  // or nullptr
  const float* coef;
  const float* frames = nullptr;  // mode T, exhaustive search: the verified search's frames of these paths, or nullptr
  // the LQ plans [P][n][2] (device-visible), candidate 2 of the LAST round, or nullptr; `before_last` - when set - runs on
  // the host right before that round is enqueued and fills them (acmpc_control_tick plans while the earlier launches
  // execute) and returns false when there is no plan after all
  const float* extra = nullptr;
  std::function<bool()> before_last;
};

// the handle's own frames (acmpc_set_paths), for the rounds that read the handle's own table
const float* own_frames(const acmpc_ctx* c) { return c->h_nn_frames.empty() ? nullptr : c->d_nn_frames; }

bool use_fused_finalize(const acmpc_ctx* c, int n) {
  return !c->sw.no_fused_finalize && acmpc::fused_finalize_fits(c->prm.mode, n);
}

// The fused finalize copies the record out of the winning workgroup's trace when the launch is small enough for the
// trace buffer (closed-loop rounds are: 256 workgroups) and the trace fits the LDS; else it re-draws and re-rolls.
bool use_traced_finalize(const acmpc_ctx* c, int P, int N, int n) {
  return !c->sw.no_traced_finalize && acmpc::traced_finalize_fits(c->prm.mode, n) &&
         static_cast<long long>(P) * ((N + 63) / 64) <= kTraceBlocks;
}

// One LQ plan (csrc/acmpc_lq.h) into `out` [n][2]: the path's 7 x n float64 table, the start state as the rollouts take it
// (mode S: the Frenet state; mode T: the pose, moved into the Frenet frame of the first waypoint here).  Without a finite
// plan (a singular step, a speed profile that was never solved) `out` gets the reference controls clipped into the box -
// candidate 1 again, harmless - and false comes back.
// With lq_candidate = 2 the plan is then refined against the QP's box rows (csrc/acmpc_lq_box.h; the iterate of problem
// `problem` is kept in the handle between calls).
bool lq_plan_into(acmpc_ctx* c, const double* table, int n, const double start[3], float* out,
                  bool start_is_pose = false, int problem = 0) {
  double x0[3] = {start[0], start[1], start[2]};
  if (start_is_pose || c->prm.mode == ACMPC_MODE_TEMPORAL) acmpc::lq::frenet_start(table, n, start, x0);
  const float lo[2] = {c->w.ulo0, c->w.ulo1}, hi[2] = {c->w.uhi0, c->w.uhi1};
  const bool finite_start = std::isfinite(x0[0]) && std::isfinite(x0[1]) && std::isfinite(x0[2]);
  if (finite_start && acmpc::lq::plan(table, n, x0, c->prm.step_cost, c->prm.r_term, c->prm.final_cost, lo, hi, out)) {
    if (c->prm.lq_candidate == 2) {
      if (c->lq_box_state.size() <= static_cast<size_t>(problem)) c->lq_box_state.resize(static_cast<size_t>(problem) + 1);
      c->lq_box_last = acmpc::lqbox::refine(table, n, x0, c->prm.step_cost, c->prm.r_term, c->prm.final_cost, lo, hi,
                                            c->prm.margin, c->prm.w_bound, c->lq_box_iterations,
                                            c->lq_box_state[static_cast<size_t>(problem)], c->lq_box_ws, out);
    }
    return true;
  }
  if (c->prm.lq_candidate == 2 && c->lq_box_state.size() > static_cast<size_t>(problem))
    c->lq_box_state[static_cast<size_t>(problem)].reset();
  const double *kappa = table + 3 * static_cast<size_t>(n), *vel = table + 6 * static_cast<size_t>(n);
  for (int i = 0; i < n; ++i) {
    out[2 * i] = std::fmin(std::fmax(static_cast<float>(vel[i]), lo[0]), hi[0]);
    out[2 * i + 1] = std::fmin(std::fmax(static_cast<float>(kappa[i]), lo[1]), hi[1]);
  }
  return false;
}

// `final_records`: where the LAST round's records go when the fused finalize writes them (device memory, or pinned
// host memory - then the winner lands in the caller's staging buffer without a copy node); nullptr = c->d_records
int enqueue_rounds(acmpc_ctx* c, const OptInputs& in, int P, int N, int n, int rounds, double sigma_v, double sigma_k,
                   double shrink, uint64_t seed, const uint32_t* d_seed, hipStream_t s, bool fused,
                   float* final_records = nullptr, unsigned* done = nullptr, unsigned done_value = 0) {
  const bool has_uref = in.uref != nullptr;
  const bool fused_finalize = use_fused_finalize(c, n);
  const int layout = ACMPC_LAYOUT_STEP_MAJOR;
  const int rec_floats = acmpc_record_floats(n);
  double scale = 1.0;
  for (int r = 0; r < rounds; ++r, scale *= shrink) {
    // round 0 samples round the caller's centre, later rounds round the incumbent = the u block of the records
    const float* d_c = (r == 0) ? in.centre : c->d_records + ACMPC_REC_HEADER;
    const int stride = (r == 0) ? 2 * n : rec_floats;
    const float* d_ref = has_uref ? in.uref : nullptr;
    const float* d_extra = nullptr;   // the LQ plan competes in the last round only
    if (r + 1 == rounds && in.extra != nullptr && (!in.before_last || in.before_last())) d_extra = in.extra;
    if (!fused) {  // (only with the handle's own buffers: in.coef == c->d_coef)
      const bool softmin = c->prm.centre_update == 1;
      // softmin rounds: candidate 0 = the weighted mean of the previous round (written into d_centre below),
      // candidate 1 = the previous round's winner, so the best plan found so far is never lost
      const bool mean_round = softmin && r > 0;
      if (mean_round)  // candidate 1 reads its controls at a stride of 2n: stage the winner's u block contiguously
        ACMPC_HIP(c, hipMemcpy2DAsync(c->d_uref, static_cast<size_t>(2 * n) * sizeof(float),
                                      c->d_records + ACMPC_REC_HEADER, static_cast<size_t>(rec_floats) * sizeof(float),
                                      static_cast<size_t>(2 * n) * sizeof(float), P, hipMemcpyDeviceToDevice, s));
      int rc = sample(c, mean_round ? c->d_centre : d_c, mean_round ? 2 * n : stride, mean_round ? c->d_uref : d_ref, P,
                      N, n, layout, 0, sigma_v * scale, sigma_k * scale, seed, static_cast<uint32_t>(r), c->d_U, s,
                      d_seed, d_extra);
      if (rc != ACMPC_OK) return rc;
      acmpc::LaunchShape shape;
      rc = rollout(c, in.x0, c->d_U, P, N, n, layout, 0, softmin ? c->d_costs : nullptr, s, &shape);
      if (rc != ACMPC_OK) return rc;
      rc = finalize(c, nullptr, softmin ? c->d_keys : nullptr, in.x0, c->d_U, P, N, n, layout, 0, c->d_records,
                    shape.blocks_per_problem, s);
      if (rc != ACMPC_OK) return rc;
      if (softmin && r + 1 < rounds) {
        acmpc::SoftminArgs sm{};
        sm.costs = c->d_costs;
        sm.keys = c->d_keys;
        sm.U = c->d_U;
        sm.partial = c->d_soft_partial;
        sm.mean = c->d_centre;   // [P][n][2]: the next round's centre
        sm.weight_sum = nullptr;
        sm.chunks = acmpc::softmin_chunks(N);
        sm.P = P;
        sm.N = N;
        sm.n = n;
        sm.lambda = static_cast<float>(c->prm.softmin_lambda);
        ACMPC_HIP(c, acmpc::launch_softmin(layout, sm, s));
      }
      continue;
    }
    int rc = upload_segments(c, n, s);
    if (rc != ACMPC_OK) return rc;
    // Traced rounds are chained: a round that is not the last ends without a finalize - its workgroups leave their
    // partial keys and the trace of their best candidate - and the NEXT launch finds the winner itself (argmin over
    // those keys while its Philox draws run) and samples round that workgroup's trace.  Only the last round pays the
    // last-workgroup tail (six dependent device-scope round trips, ~10 us).  Keys, counts and traces alternate between
    // two sets, since a round reads its predecessor's while it writes its own.
    const int blocks = (N + 63) / 64;
    const bool traced = fused_finalize && use_traced_finalize(c, P, N, n);
    const bool chain = traced && blocks <= acmpc::kChainBlocks && !c->sw.no_chained_rounds;
    const size_t set = (chain && (r & 1)) ? 1 : 0;
    const size_t trace_set_floats = static_cast<size_t>(kTraceBlocks) * acmpc::trace_floats(c->prm.max_steps);
    float* d_trace = c->d_trace + set * trace_set_floats;
    acmpc::RolloutArgs ra{};
    ra.x0 = in.x0;
    ra.coef = in.coef;
    ra.nn_frames = !c->sw.no_verified_search ? in.frames : nullptr;
    ra.partial_keys = c->d_partial_keys + set * c->partial_slots;
    ra.partial_feas = c->d_partial_feas + set * c->partial_slots;
    ra.P = P;
    ra.N = N;
    ra.n = n;
    ra.index_offset = 0;
    ra.w = c->w;
    acmpc::SampleArgs sa{};
    sa.centre = d_c;
    sa.centre_stride = stride;
    sa.u_ref = d_ref;
    sa.u_extra = d_extra;
    sa.P = P;
    sa.N = N;
    sa.n = n;
    sa.spec = make_spec(c, sigma_v * scale, sigma_k * scale, seed, static_cast<uint32_t>(r));
    sa.spec.seed_ptr = d_seed;
    if (chain && r > 0) {
      sa.prev_keys = c->d_partial_keys + (set ^ 1) * c->partial_slots;
      sa.prev_trace = c->d_trace + (set ^ 1) * trace_set_floats;
      sa.prev_blocks = blocks;
      sa.prev_pitch = acmpc::trace_floats(n);
    }
    // NB: the finalize of round r reads its centre from the records it is about to overwrite; it copies the
    // controls it needs into registers/LDS before lane 0..63 write the new record, and one wave owns one record
    // (timing armed - acmpc_profile_enable, eager path only: every round's launch carries an event pair)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof_used < c->prof_start.size()) {
      e0 = c->prof_start[c->prof_used];
      e1 = c->prof_stop[c->prof_used];
      ++c->prof_used;
    }
    if (fused_finalize) {
      // one launch per round: the last workgroup of each problem also reduces the partial keys and writes the
      // record; rounds before the last only need the winner's controls (the next centre), not its re-roll
      const bool last = r + 1 == rounds;
      const bool tail = last || !chain;
      const acmpc::FusedFinalize ff{tail ? c->d_tickets : nullptr,
                                    (last && final_records != nullptr) ? final_records : c->d_records, !last,
                                    traced ? d_trace : nullptr, acmpc::trace_floats(n),
                                    last ? done : nullptr, done_value};
      ACMPC_HIP(c, acmpc::launch_rollout_sampled(c->prm.mode, ra, sa, ff, s, e0, e1, c->opt));
    } else {
      ACMPC_HIP(c, acmpc::launch_rollout_sampled(c->prm.mode, ra, sa, acmpc::FusedFinalize{nullptr, nullptr, false, nullptr, 0, nullptr, 0}, s, e0, e1, c->opt));
      Regenerate regen{d_c, stride, d_ref, sa.spec, d_extra};
      rc = finalize(c, nullptr, nullptr, in.x0, nullptr, P, N, n, layout, 0, c->d_records, (N + 63) / 64, s, &regen,
                    in.coef);
      if (rc != ACMPC_OK) return rc;
    }
  }
  return ACMPC_OK;
}

}  // namespace
}  // extern "C++"

int acmpc_optimize(acmpc_ctx* c, const float* x0, const float* centre, const float* u_ref, int32_t P, int32_t N,
                   int32_t n, int32_t rounds, const double sigma[2], double shrink, uint64_t seed, float* records) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (x0 == nullptr || centre == nullptr || sigma == nullptr || records == nullptr)
    return fail(c, ACMPC_EINVAL, "null argument");
  if (rounds < 1) return fail(c, ACMPC_EINVAL, "rounds must be positive");
  const int layout = ACMPC_LAYOUT_STEP_MAJOR;
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_staging(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = c->stream;
  const size_t x0_bytes = static_cast<size_t>(P) * 3 * sizeof(float);
  const size_t path_bytes = static_cast<size_t>(P) * n * 2 * sizeof(float);
  const size_t table_bytes = static_cast<size_t>(P) * n * c->coef_stride * sizeof(float);
  const size_t rec_bytes = static_cast<size_t>(P) * acmpc_record_floats(n) * sizeof(float);
  const bool has_uref = u_ref != nullptr;
  // the LQ plans of these paths from these start states: candidate 2 of the last round (acmpc_params::lq_candidate)
  const bool has_extra = c->prm.lq_candidate != 0 && c->h_tables.size() == static_cast<size_t>(P) * 7 * n;
  if (has_extra) {
    for (int p = 0; p < P; ++p) {
      const double start[3] = {x0[3 * p], x0[3 * p + 1], x0[3 * p + 2]};
      (void)lq_plan_into(c, c->h_tables.data() + static_cast<size_t>(p) * 7 * n, n, start, c->h_lq + static_cast<size_t>(p) * n * 2,
                         false, p);
    }
  }

  // Eager path: when rollout launches are being timed (event pairs cannot be captured) or on request.
  if (c->prof_used < c->prof_start.size() || c->sw.no_graph) {
    rc = upload_tables(c, s);
    if (rc != ACMPC_OK) return rc;
    ACMPC_HIP(c, hipMemcpyAsync(c->d_x0, x0, x0_bytes, hipMemcpyHostToDevice, s));
    ACMPC_HIP(c, hipMemcpyAsync(c->d_centre, centre, path_bytes, hipMemcpyHostToDevice, s));
    if (has_uref) ACMPC_HIP(c, hipMemcpyAsync(c->d_uref, u_ref, path_bytes, hipMemcpyHostToDevice, s));
    OptInputs in{c->d_x0, c->d_centre, has_uref ? c->d_uref : nullptr, c->d_coef, own_frames(c)};
    in.extra = has_extra ? c->h_lq : nullptr;   // (pinned: the last round reads the plans in place)
    rc = enqueue_rounds(c, in, P, N, n, rounds, sigma[0], sigma[1], shrink, seed, nullptr, s,
                        !c->sw.no_fused_sampling && c->prm.centre_update == 0);
    if (rc != ACMPC_OK) return rc;
    ACMPC_HIP(c, hipMemcpyAsync(records, c->d_records, rec_bytes, hipMemcpyDeviceToHost, s));
    ACMPC_HIP(c, hipStreamSynchronize(s));
    return ACMPC_OK;
  }

  // Graph path.  Pinned staging block layout: x0 | centre | u_ref | table | seed (each 16-byte aligned).
  auto align16 = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
  const size_t off_x0 = 0, off_centre = align16(off_x0 + x0_bytes), off_uref = align16(off_centre + path_bytes),
               off_table = align16(off_uref + path_bytes), off_seed = align16(off_table + table_bytes);
  if (!c->opt_ready) {
    const acmpc_params& p = c->prm;
    const size_t cap = 64 + 16 * 5 + static_cast<size_t>(p.max_problems) *
                                         (3 + 4 * static_cast<size_t>(p.max_steps) +
                                          static_cast<size_t>(p.max_steps) * c->coef_stride) * sizeof(float);
    ACMPC_HIP(c, host_alloc_once(&c->h_opt, cap));
    ACMPC_HIP(c, alloc_once(&c->d_opt, cap));
    c->opt_capacity = cap;
    ACMPC_HIP(c, host_alloc_once(&c->h_opt_records,
                                 static_cast<size_t>(p.max_problems) * acmpc_record_floats(p.max_steps) * sizeof(float)));
    ACMPC_HIP(c, alloc_once(&c->d_seed, 2 * sizeof(uint32_t)));
    c->opt_ready = true;
  }
  acmpc_ctx::OptKey key;
  key.P = P;
  key.N = N;
  key.n = n;
  key.rounds = rounds;
  key.has_uref = (has_uref ? 1 : 0) | (has_extra ? 2 : 0);
  key.sigma_v = sigma[0];
  key.sigma_k = sigma[1];
  key.shrink = shrink;
  int slot = -1;
  for (int g = 0; g < acmpc_ctx::kOptGraphs; ++g)
    if (c->opt_graph[g] != nullptr && key == c->opt_key[g]) slot = g;
  if (slot < 0) {
    slot = 0;
    for (int g = 1; g < acmpc_ctx::kOptGraphs; ++g)
      if (c->opt_used[g] < c->opt_used[slot]) slot = g;
    if (c->opt_graph[slot] != nullptr) {
      (void)hipGraphExecDestroy(c->opt_graph[slot]);
      c->opt_graph[slot] = nullptr;
    }
    rc = upload_segments(c, n, s);  // must not happen inside the capture (it synchronises)
    if (rc != ACMPC_OK) return rc;
    hipGraph_t graph = nullptr;
    ACMPC_HIP(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    // ONE host-to-device copy brings x0, centre, u_ref, the table and the seed; the kernels read them in place
    const size_t in_bytes = off_seed + 2 * sizeof(uint32_t);
    hipError_t e = hipMemcpyAsync(c->d_opt, c->h_opt, in_bytes, hipMemcpyHostToDevice, s);
    const bool fused = !c->sw.no_fused_sampling && c->prm.centre_update == 0;
    OptInputs in{reinterpret_cast<const float*>(c->d_opt + off_x0), reinterpret_cast<const float*>(c->d_opt + off_centre),
                 has_uref ? reinterpret_cast<const float*>(c->d_opt + off_uref) : nullptr,
                 reinterpret_cast<const float*>(c->d_opt + off_table), own_frames(c)};
    in.extra = has_extra ? c->h_lq : nullptr;
    if (!fused) {  // the three-kernel form runs on the handle's own buffers: copy the block's parts there
      auto spread = [&](void* dst, size_t off, size_t bytes) {
        if (e == hipSuccess) e = hipMemcpyAsync(dst, c->d_opt + off, bytes, hipMemcpyDeviceToDevice, s);
      };
      spread(c->d_x0, off_x0, x0_bytes);
      spread(c->d_centre, off_centre, path_bytes);
      if (has_uref) spread(c->d_uref, off_uref, path_bytes);
      spread(c->d_coef, off_table, table_bytes);
      in = OptInputs{c->d_x0, c->d_centre, has_uref ? c->d_uref : nullptr, c->d_coef, own_frames(c)};
      in.extra = has_extra ? c->h_lq : nullptr;
    }
    int rc_rounds = ACMPC_OK;
    // with the fused finalize the last round writes the winners straight into the pinned host buffer (posted
    // writes over the host link, visible once the stream has drained): no device-to-host copy node
    const bool direct = fused && use_fused_finalize(c, n);
    if (e == hipSuccess)
      rc_rounds = enqueue_rounds(c, in, P, N, n, rounds, sigma[0], sigma[1], shrink, 0,
                                 reinterpret_cast<const uint32_t*>(c->d_opt + off_seed), s, fused,
                                 direct ? c->h_opt_records : nullptr);
    if (e == hipSuccess && rc_rounds == ACMPC_OK && !direct)
      e = hipMemcpyAsync(c->h_opt_records, c->d_records, rec_bytes, hipMemcpyDeviceToHost, s);
    const hipError_t e_end = hipStreamEndCapture(s, &graph);
    if (rc_rounds != ACMPC_OK) {
      if (graph != nullptr) (void)hipGraphDestroy(graph);
      return rc_rounds;
    }
    if (e != hipSuccess) {
      if (graph != nullptr) (void)hipGraphDestroy(graph);
      return fail_hip(c, e, "capturing the optimisation graph");
    }
    ACMPC_HIP(c, e_end);
    const hipError_t e_inst = hipGraphInstantiate(&c->opt_graph[slot], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e_inst != hipSuccess) c->opt_graph[slot] = nullptr;
    ACMPC_HIP(c, e_inst);
    c->opt_key[slot] = key;
  }
  c->opt_used[slot] = ++c->opt_clock;
  std::memcpy(c->h_opt + off_x0, x0, x0_bytes);
  std::memcpy(c->h_opt + off_centre, centre, path_bytes);
  if (has_uref) std::memcpy(c->h_opt + off_uref, u_ref, path_bytes);
  std::memcpy(c->h_opt + off_table, c->h_coef.data(), table_bytes);
  const uint32_t seed_words[2] = {static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)};
  std::memcpy(c->h_opt + off_seed, seed_words, sizeof seed_words);
  rc = upload_frames(c, s);
  if (rc != ACMPC_OK) return rc;
  ACMPC_HIP(c, hipGraphLaunch(c->opt_graph[slot], s));
  ACMPC_HIP(c, hipStreamSynchronize(s));
  std::memcpy(records, c->h_opt_records, rec_bytes);
  return ACMPC_OK;
}


extern "C++" {
namespace {

size_t align16(size_t v) { return (v + 15) & ~static_cast<size_t>(15); }

// layout of the tick blocks for a horizon of n steps: the pinned host input block (header | coords | centre) and the
// device block the prologue fills for the rollout kernels (seed | x0 | centre | u_ref | table)
struct TickLayout {
  size_t coords, centre_in, host_total;            // pinned host block
  size_t seed, x0, centre, uref, coef, frames, total;   // device block
  explicit TickLayout(int n, int coef_stride = ACMPC_COEF_STRIDE_SPATIAL) {
    coords = align16(sizeof(acmpc::TickHeader));
    centre_in = align16(coords + static_cast<size_t>(n + 1) * 3 * sizeof(double));
    host_total = align16(centre_in + static_cast<size_t>(n) * 2 * sizeof(float));
    seed = 0;
    x0 = 16;
    centre = 32;
    uref = align16(centre + static_cast<size_t>(n) * 2 * sizeof(float));
    coef = align16(uref + static_cast<size_t>(n) * 2 * sizeof(float));
    frames = align16(coef + static_cast<size_t>(n) * coef_stride * sizeof(float));   // (mode T, exhaustive search)
    total = align16(frames + static_cast<size_t>(acmpc::verified_frame_floats(std::max(n, acmpc::kVerifiedWindow))) * sizeof(float));
  }
};

// layout of the pinned result block
struct TickOutLayout {
  size_t record, table, status, coords, done, total;
  explicit TickOutLayout(int n) {
    record = 0;
    table = align16(static_cast<size_t>(acmpc_record_floats(n)) * sizeof(float));
    status = align16(table + static_cast<size_t>(7) * n * sizeof(double));   // QP status, iterations, map index
    coords = status + 16;
    done = align16(coords + static_cast<size_t>(n + 1) * 3 * sizeof(double));   // completion flag of the last round
    total = done + 16;
  }
};

// the bound map -> device (when bound or re-bound since the last upload)
int upload_map(acmpc_ctx* c, hipStream_t s) {
  if (!c->map_dirty) return ACMPC_OK;
  // a captured tick graph has the map's address, length and window size in its kernel arguments: none survives a re-bind
  for (hipGraphExec_t& g : c->tick_graph) {
    if (g != nullptr) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
  ACMPC_HIP(c, hipStreamSynchronize(s));   // nothing of an earlier tick still reads the old map
  (void)hipFree(c->d_map);
  c->d_map = nullptr;
  ACMPC_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_map), c->h_map.size() * sizeof(double)));
  ACMPC_HIP(c, hipMemcpyAsync(c->d_map, c->h_map.data(), c->h_map.size() * sizeof(double), hipMemcpyHostToDevice, s));
  ACMPC_HIP(c, hipStreamSynchronize(s));
  c->map_dirty = false;
  return ACMPC_OK;
}

int map_window_args(acmpc_ctx* c, int H, int points, const TickOutLayout& out, acmpc::MapWindowArgs* a) {
  if (c->h_map.empty()) return fail(c, ACMPC_ESTATE, "no map bound (acmpc_bind_map)");
  if (points < H || points % H != 0) return fail(c, ACMPC_EINVAL, "centreline_points must be a multiple of the horizon");
  a->header = reinterpret_cast<const acmpc::TickHeader*>(c->h_tick);
  a->centre = c->d_map;
  a->M = static_cast<int>(c->h_map.size() / 2);
  a->count = static_cast<int>(std::lround(150.0 / c->map_spacing)) + 1;   // BEV look-ahead, perception/tracks.py:14
  if (a->count < 2 || a->count > a->M) return fail(c, ACMPC_EINVAL, "the map is shorter than the 150 m look-ahead window");
  a->points = points;
  a->H = H;
  a->coords = c->d_coords;
  a->coords_out = reinterpret_cast<double*>(c->h_tick_out + out.coords);
  a->first_out = reinterpret_cast<int*>(c->h_tick_out + out.status) + 2;
  return ACMPC_OK;
}

int ensure_tick(acmpc_ctx* c) {
  if (c->tick_ready) return ACMPC_OK;
  c->touched_device = true;
  const int n_cap = std::min(c->prm.max_steps, acmpc::kPrologueMaxSteps);
  ACMPC_HIP(c, host_alloc_once(&c->h_tick, TickLayout(n_cap).host_total));
  ACMPC_HIP(c, alloc_once(&c->d_tick, TickLayout(n_cap).total));
  ACMPC_HIP(c, host_alloc_once(&c->h_tick_out, TickOutLayout(n_cap).total));
  std::memset(c->h_tick_out, 0, TickOutLayout(n_cap).total);   // completion flags start below every sequence number
  ACMPC_HIP(c, alloc_once(&c->d_coords, static_cast<size_t>(n_cap + 1) * 3 * sizeof(double)));
  c->warm_stride = 2 + 3 * n_cap;
  const size_t warm_bytes = static_cast<size_t>(2) * c->warm_stride * sizeof(double);
  ACMPC_HIP(c, alloc_once(&c->d_warm, warm_bytes));
  ACMPC_HIP(c, hipMemset(c->d_warm, 0, warm_bytes));  // valid flags 0: the first tick of each solver starts cold
  ACMPC_HIP(c, hipStreamSynchronize(nullptr));
  c->tick_ready = true;
  return ACMPC_OK;
}

}  // namespace
}  // extern "C++"

int acmpc_control_tick(acmpc_ctx* c, const acmpc_tick* t, const double* coords, const float* centre, double* table,
                       float* record, double* decision, double* projected_control, double* prediction,
                       double* cum_time, double* times, double* accelerations, double* steer_rates, double* info,
                       double* coords_out) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (t == nullptr || table == nullptr || record == nullptr || decision == nullptr ||
      projected_control == nullptr || prediction == nullptr || cum_time == nullptr || times == nullptr ||
      accelerations == nullptr || steer_rates == nullptr || info == nullptr)
    return fail(c, ACMPC_EINVAL, "null argument");
  if (t->struct_size != sizeof(acmpc_tick)) return fail(c, ACMPC_EINVAL, "acmpc_tick size mismatch");
  if (c->stream_pending)
    return fail(c, ACMPC_ESTATE, "a batch of acmpc_solve_stream_device is pending: acmpc_solve_stream_flush first");
  if (c->prm.centre_update != 0) return fail(c, ACMPC_ESTATE, "acmpc_control_tick needs a handle with centre_update = 0");
  const bool temporal = c->prm.mode == ACMPC_MODE_TEMPORAL;
  if (temporal && !(c->prm.dt > 0.0)) return fail(c, ACMPC_ESTATE, "mode T needs a positive dt");
  const int H = t->horizon, n = H - 1, N = t->n_candidates;
  if (H < 3 || t->rounds < 1 || N < 1) return fail(c, ACMPC_EINVAL, "need horizon >= 3, rounds >= 1, n_candidates >= 1");
  if (n > c->prm.max_steps || N > c->prm.max_candidates) return fail(c, ACMPC_ECAPACITY, "horizon or candidates exceed capacity");
  if (n > acmpc::kPrologueMaxSteps) return fail(c, ACMPC_ESTATE, "the device prologue holds at most 128 steps");
  if (centre == nullptr && t->centre_is_reference == 0) return fail(c, ACMPC_EINVAL, "null centre");
  if (coords == nullptr && c->h_map.empty()) return fail(c, ACMPC_EINVAL, "null coords and no map bound");
  if (!acmpc::fused_finalize_fits(c->prm.mode, n)) return fail(c, ACMPC_ESTATE, "fused finalize does not fit");
  int rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_staging(c);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_tick(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = c->stream;
  const TickLayout in(n, c->coef_stride);   // (spatial rows are the wider: ensure_tick sized the blocks for them)
  const TickOutLayout out(n);
  const int rec_floats = acmpc_record_floats(n);
  const bool from_map = coords == nullptr;
  acmpc::MapWindowArgs ma{};
  if (from_map) {
    rc = map_window_args(c, H, t->centreline_points, out, &ma);
    if (rc != ACMPC_OK) return rc;
    rc = upload_map(c, s);
    if (rc != ACMPC_OK) return rc;
    ma.centre = c->d_map;
  }

  acmpc::PrologueArgs pa{};
  pa.header = reinterpret_cast<const acmpc::TickHeader*>(c->h_tick);   // read in place over the host link
  pa.coords = reinterpret_cast<const double*>(c->h_tick + in.coords);
  if (from_map) {
    pa.map_centre = c->d_map;
    pa.map_M = ma.M;
    pa.map_count = ma.count;
    pa.map_points = ma.points;
    pa.map_first = reinterpret_cast<const int*>(c->d_coords);   // (the search kernel leaves the index here)
    pa.coords_out = ma.coords_out;
    pa.index_out = ma.first_out;
    ma.coords = nullptr;                                         // the search launch only needs to leave `first`
    ma.coords_out = nullptr;
    ma.first_out = reinterpret_cast<int*>(c->d_coords);
  }
  pa.temporal = temporal ? 1 : 0;
  pa.centre_in = reinterpret_cast<const float*>(c->h_tick + in.centre_in);
  pa.x0 = reinterpret_cast<float*>(c->d_tick + in.x0);
  pa.u_ref = reinterpret_cast<float*>(c->d_tick + in.uref);
  pa.coef = reinterpret_cast<float*>(c->d_tick + in.coef);
  // the frames of the verified search: tabulated (by the prologue's second workgroup) only when the rounds can take them
  // - beyond 106 steps they no longer fit the three-wave round's LDS and the search wave scans every waypoint
  pa.frames = (temporal && c->prm.nn_ahead < 0 && n >= acmpc::kVerifiedWindow && acmpc::trio_frames_fit(n) &&
               !c->opt.no_trio_rounds && !c->sw.no_verified_search)
                  ? reinterpret_cast<float*>(c->d_tick + in.frames)
                  : nullptr;
  pa.centre = reinterpret_cast<float*>(c->d_tick + in.centre);
  pa.seed = reinterpret_cast<uint32_t*>(c->d_tick + in.seed);
  pa.table_out = reinterpret_cast<double*>(c->h_tick_out + out.table);
  pa.status = reinterpret_cast<int*>(c->h_tick_out + out.status);
  pa.warm_state = c->d_warm;
  pa.warm_stride = c->warm_stride;
  pa.warm_capacity = (c->warm_stride - 2) / 3;
  pa.margin = c->prm.margin;
  pa.u_lo0 = c->prm.u_min[0];
  pa.u_lo1 = c->prm.u_min[1];
  pa.u_hi0 = c->prm.u_max[0];
  pa.u_hi1 = c->prm.u_max[1];
  const bool direct = use_fused_finalize(c, n);
  // Completion: with direct launches the last round's tail stores a sequence number behind the record, both in pinned
  // host memory, and this call polls it - the record is here a microsecond after it was written, where the launch's
  // completion signal (hipStreamSynchronize) takes the driver's path.  The stream is only synchronised when the flag
  // does not come (a fault), and before a host buffer the kernels read is rewritten by a DIFFERENT kind of call.
  const bool use_graph = c->sw.tick_graph;
  // The LQ plan (acmpc_params::lq_candidate), computed on the host while this tick's prologue and earlier rounds run and
  // read by the last round in place from pinned memory.  The tick's own table is being built on the device right now; what
  // the host has is this tick's PATH - so the plan is for the waypoints of `coords` (acmpc_waypoint_table, the host
  // statement of the prologue's first step) with the speed profile the previous tick solved (the QP is warm-started from
  // it and moves little from tick to tick) and the start state of this tick's pose (offset, 0, pi / 2).  With the path cut
  // out of the map on the device (coords = NULL) the host does not have it: the plan is then the previous tick's problem's.
  const bool lq_on = c->prm.lq_candidate != 0;
  const double lq_offset = t->offset;
  // With the path cut out of the map on the device the host cuts the same window itself (round 5) when it knows where it
  // starts - `map_index` given; for a pose, whose nearest map point the device searches, the plan stays the previous tick's
  // problem's: a scan of the map on the host would outlast the rounds it has to hide behind.
  const bool host_window = lq_on && from_map && t->map_index >= 0;
  const int window_M = ma.M, window_count = ma.count, window_points = ma.points;
  auto plan_previous = [c, n, H, given = coords, lq_offset, t, host_window, window_M, window_count, window_points]() -> bool {
    const double* coords = given;
    if (host_window) {   // (here, not in front of the launches: this runs while the prologue and the first round do)
      c->tick_host_coords.resize(static_cast<size_t>(H) * 3);
      const int first = ((t->map_index % window_M) + window_M) % window_M;
      const acmpc::MapFrame frame = acmpc::map_frame(c->h_map.data(), window_M, first);
      for (int r = 0; r < H; ++r) {
        double row[3];
        acmpc::map_path_row(c->h_map.data(), window_M, first, window_count, window_points, H, r, t->lateral_offset, frame, row);
        for (int e = 0; e < 3; ++e) c->tick_host_coords[static_cast<size_t>(3) * r + e] = row[e];
      }
      coords = c->tick_host_coords.data();
    }
    // The speed profile the plan is made with.  With the path on the host and the exact profile (qp_method 0) it is THIS
    // tick's - the host statement of the prologue's own two passes (acmpc_velocity_ceiling + acmpc_speed_profile_exact, a
    // microsecond) - on the host's waypoint table; where that does not apply (an infeasible profile; qp_method 1: the
    // splitting is not run twice per tick) the previous tick's, and with no previous tick either (a handle's first, another
    // horizon, a tick without a finite plan) the splitting, cold, once.
    const bool have_previous = c->tick_prev_n == n;
    if (!have_previous && coords == nullptr) return false;   // (no path on the host: nothing to plan for)
    if (coords == nullptr) return lq_plan_into(c, c->tick_prev_table.data(), n, c->tick_prev_x0, c->h_lq);
    c->tick_lq_table.resize(static_cast<size_t>(7) * n);
    if (acmpc_waypoint_table(coords, H, kEps, c->tick_lq_table.data()) != ACMPC_OK) return false;
    {
      c->tick_lq_scratch.resize(static_cast<size_t>(3) * n);
      double* ceiling = c->tick_lq_scratch.data();
      double* dual = ceiling + n;
      double* profile = c->tick_lq_table.data() + static_cast<size_t>(6) * n;
      const double* spacing = c->tick_lq_table.data() + static_cast<size_t>(4) * n;
      int32_t iterations = 0;
      if (acmpc_velocity_ceiling(c->tick_lq_table.data() + static_cast<size_t>(3) * n, n, t->ay_max, t->ki_min, t->v_min,
                                 t->v_max, t->localised, t->has_end_velocity, t->end_velocity, ceiling) != ACMPC_OK)
        return false;
      const bool swept = t->qp_method == 0 &&
                         acmpc_speed_profile_exact(ceiling, spacing, n, t->a_min, t->a_max, t->v_min, profile, dual) == 0;
      if (!swept) {
        if (have_previous) {
          std::memcpy(profile, c->tick_prev_table.data() + static_cast<size_t>(6) * n, static_cast<size_t>(n) * sizeof(double));
        } else if (acmpc_speed_profile_qp(ceiling, spacing, n, t->a_min, t->a_max, t->v_min, t->qp_max_iter, t->qp_check_every,
                                          t->qp_eps_abs, t->qp_eps_rel, profile, dual, 0, &iterations) != 0) {
          return false;
        }
      }
    }
    const double pose[3] = {lq_offset, 0.0, M_PI / 2.0};
    return lq_plan_into(c, c->tick_lq_table.data(), n, pose, c->h_lq, true);
  };
  const bool flagged = direct && !use_graph && !c->sw.tick_no_flag;
  unsigned* done_flag = reinterpret_cast<unsigned*>(c->h_tick_out + out.done);
  const unsigned done_value = ++c->tick_sequence;
  // prologue -> rounds (-> copy of the record when the fused finalize cannot write it to the host itself)
  auto enqueue = [&](hipStream_t q, int* rc_rounds) -> hipError_t {
    // (a pose instead of a map index: the nearest-point search runs in front, as its own 256-thread launch)
    hipError_t e = (from_map && t->map_index < 0) ? acmpc::launch_map_window(ma, q) : hipSuccess;
    if (e == hipSuccess) e = acmpc::launch_prologue(pa, n, q);
    if (e != hipSuccess) return e;
    OptInputs oi{pa.x0, pa.centre, pa.u_ref, pa.coef, pa.frames};
    if (lq_on) {
      oi.extra = c->h_lq;
      if (!use_graph) oi.before_last = plan_previous;   // (a captured graph: planned before the replay, below)
    }
    // (launched directly the rounds take the seed by value: read from the device block, as a replayed graph must, it is
    // a dependent load in front of every round's first Philox draw)
    *rc_rounds = enqueue_rounds(c, oi, 1, N, n, t->rounds, t->sigma[0], t->sigma[1], t->shrink,
                                use_graph ? 0 : t->seed, use_graph ? pa.seed : nullptr, q, true,
                                direct ? reinterpret_cast<float*>(c->h_tick_out + out.record) : nullptr,
                                flagged ? done_flag : nullptr, done_value);
    if (*rc_rounds == ACMPC_OK && !direct)
      e = hipMemcpyAsync(c->h_tick_out + out.record, c->d_records, static_cast<size_t>(rec_floats) * sizeof(float),
                         hipMemcpyDeviceToHost, q);
    return e;
  };
  // Three short kernels behind one another: launched directly they start sooner than a graph replay does (the
  // replay's fixed cost is ~10 us on this runtime, a launch on an idle stream ~4 us, and the later launches overlap
  // the prologue's execution).  ACMPC_TICK_GRAPH=1 replays a captured graph instead.
  int slot = -1;
  if (use_graph) {
    acmpc_ctx::TickKey key;
    key.N = N;
    key.n = n;
    key.rounds = t->rounds;
    key.sigma_v = t->sigma[0];
    key.sigma_k = t->sigma[1];
    key.shrink = t->shrink;
    key.from_map = from_map ? (t->map_index < 0 ? -t->centreline_points : t->centreline_points) : 0;
    for (int g = 0; g < acmpc_ctx::kOptGraphs; ++g)
      if (c->tick_graph[g] != nullptr && key == c->tick_key[g]) slot = g;
    if (slot < 0) {
      slot = 0;
      for (int g = 1; g < acmpc_ctx::kOptGraphs; ++g)
        if (c->tick_used[g] < c->tick_used[slot]) slot = g;
      if (c->tick_graph[slot] != nullptr) {
        (void)hipGraphExecDestroy(c->tick_graph[slot]);
        c->tick_graph[slot] = nullptr;
      }
      rc = upload_segments(c, n, s);  // must not happen inside the capture (it synchronises)
      if (rc != ACMPC_OK) return rc;
      hipGraph_t graph = nullptr;
      ACMPC_HIP(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int rc_rounds = ACMPC_OK;
      const hipError_t e = enqueue(s, &rc_rounds);
      const hipError_t e_end = hipStreamEndCapture(s, &graph);
      if (rc_rounds != ACMPC_OK) {
        if (graph != nullptr) (void)hipGraphDestroy(graph);
        return rc_rounds;
      }
      if (e != hipSuccess) {
        if (graph != nullptr) (void)hipGraphDestroy(graph);
        return fail_hip(c, e, "capturing the tick graph");
      }
      ACMPC_HIP(c, e_end);
      const hipError_t e_inst = hipGraphInstantiate(&c->tick_graph[slot], graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e_inst != hipSuccess) c->tick_graph[slot] = nullptr;
      ACMPC_HIP(c, e_inst);
      c->tick_key[slot] = key;
    }
    c->tick_used[slot] = ++c->opt_clock;
  } else {
    rc = upload_segments(c, n, s);  // (a no-op once the table for this n is resident)
    if (rc != ACMPC_OK) return rc;
  }

  acmpc::TickHeader* h = reinterpret_cast<acmpc::TickHeader*>(c->h_tick);
  h->offset = t->offset;
  h->v_min = t->v_min;
  h->v_max = t->v_max;
  h->a_min = t->a_min;
  h->a_max = t->a_max;
  h->ay_max = t->ay_max;
  h->ki_min = t->ki_min;
  h->end_velocity = t->end_velocity;
  h->qp_eps_abs = t->qp_eps_abs;
  h->qp_eps_rel = t->qp_eps_rel;
  h->eps = kEps;
  h->horizon = H;
  h->localised = t->localised;
  h->has_end_velocity = t->has_end_velocity;
  h->centre_is_reference = t->centre_is_reference;
  h->qp_max_iter = t->qp_max_iter;
  h->qp_check_every = t->qp_check_every;
  h->qp_method = t->qp_method;
  h->seed_lo = static_cast<uint32_t>(t->seed);
  h->seed_hi = static_cast<uint32_t>(t->seed >> 32);
  h->use_map = from_map ? 1 : 0;
  h->map_index = t->map_index;
  h->pose_x = t->pose_x;
  h->pose_y = t->pose_y;
  h->lateral_offset = t->lateral_offset;
  if (!from_map) std::memcpy(c->h_tick + in.coords, coords, static_cast<size_t>(H) * 3 * sizeof(double));
  if (centre != nullptr) std::memcpy(c->h_tick + in.centre_in, centre, static_cast<size_t>(n) * 2 * sizeof(float));
  if (use_graph) {
    if (lq_on && !plan_previous()) {
      // no plan for the replayed graph's candidate 2, which always reads h_lq: it gets the centre sequence instead
      // (candidate 0 again) or, without one, zeros - the sampler clips them into the input box like every candidate, so
      // the slot holds a DEFINED sequence (never the stale plan of another path) that the argmin will not keep
      if (centre != nullptr && t->centre_is_reference == 0)
        std::memcpy(c->h_lq, centre, static_cast<size_t>(n) * 2 * sizeof(float));
      else
        std::memset(c->h_lq, 0, static_cast<size_t>(n) * 2 * sizeof(float));
    }
    ACMPC_HIP(c, hipGraphLaunch(c->tick_graph[slot], s));
  } else {
    pa.header_by_value = 1;
    pa.header_value = *h;
    if (!from_map && H <= acmpc::kInlinePathPoints && !c->sw.tick_no_inline_path) {
      pa.path_by_value = 1;
      std::memcpy(pa.coords_value, coords, static_cast<size_t>(H) * 3 * sizeof(double));
      if (centre != nullptr) std::memcpy(pa.centre_value, centre, static_cast<size_t>(n) * 2 * sizeof(float));
    }
    int rc_rounds = ACMPC_OK;
    const hipError_t e = enqueue(s, &rc_rounds);
    if (rc_rounds != ACMPC_OK || e != hipSuccess) {
      // part of the sequence may be running: it reads the pinned input block and writes the result block, which the
      // caller's next tick would overwrite - wait for it (result ignored in favour of the error that brought us here)
      (void)hipStreamSynchronize(s);
      if (rc_rounds != ACMPC_OK) return rc_rounds;
      ACMPC_HIP(c, e);
    }
  }
  if (flagged) {
    volatile unsigned* flag = done_flag;
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
    unsigned spins = 0;
    while (*flag != done_value) {
      __builtin_ia32_pause();
      if ((++spins & 0x3fffu) == 0 && std::chrono::steady_clock::now() > give_up) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag != done_value) {   // no flag: wait the ordinary way, which also reports what went wrong
      ACMPC_HIP(c, hipStreamSynchronize(s));
      if (*flag != done_value) return fail(c, ACMPC_EHIP, "the tick finished without its completion flag");
    }
  } else {
    ACMPC_HIP(c, hipStreamSynchronize(s));
  }
  c->tick_last_n = n;

  const float* rec = reinterpret_cast<const float*>(c->h_tick_out + out.record);
  std::memcpy(record, rec, static_cast<size_t>(rec_floats) * sizeof(float));
  std::memcpy(table, c->h_tick_out + out.table, static_cast<size_t>(7) * n * sizeof(double));
  const int* status = reinterpret_cast<const int*>(c->h_tick_out + out.status);
  // dec.x = [x_0 .. x_n ; u_0 .. u_{n-1}] (control.py:121-158) from the record's [u ; x] blocks
  const float* ru = rec + ACMPC_REC_HEADER;
  const float* rx = ru + 2 * n;
  double biggest = 0.0;
  bool finite = std::isfinite(rec[ACMPC_REC_COST]) && std::isfinite(rec[ACMPC_REC_VIOLATION]);
  for (int i = 0; i < 3 * (n + 1); ++i) {
    decision[i] = static_cast<double>(rx[i]);
    biggest = std::max(biggest, std::fabs(decision[i]));
    finite = finite && std::isfinite(rx[i]);
  }
  for (int i = 0; i < 2 * n; ++i) {
    decision[3 * (n + 1) + i] = static_cast<double>(ru[i]);
    biggest = std::max(biggest, std::fabs(static_cast<double>(ru[i])));
    finite = finite && std::isfinite(ru[i]);
  }
  rc = temporal ? acmpc_unpack_decision_temporal(decision, n, c->prm.dt, c->prm.wheelbase, projected_control, prediction,
                                                 cum_time, times, accelerations, steer_rates)
                : acmpc_unpack_decision(decision, n, table, c->prm.wheelbase, projected_control, prediction, cum_time,
                                        times, accelerations, steer_rates);
  if (rc != ACMPC_OK) return fail(c, rc, "acmpc_unpack_decision");
  info[0] = rec[ACMPC_REC_COST];
  info[1] = rec[ACMPC_REC_VIOLATION];
  info[2] = rec[ACMPC_REC_NFEASIBLE];
  info[3] = biggest;
  info[4] = status[0];
  info[5] = status[1];
  info[6] = from_map ? static_cast<double>(status[2]) : -1.0;   // first map index of the window
  info[7] = finite ? 0.0 : 1.0;   // a non-finite cost, violation or plan entry (max |dec.x| above skips NaNs)
  if (lq_on) {   // what the next tick plans for: this tick's table and start state (the record's x_0: Frenet state or pose)
    c->tick_prev_n = (finite && status[0] == 0) ? n : 0;
    if (c->tick_prev_n != 0) {
      c->tick_prev_table.assign(table, table + static_cast<size_t>(7) * n);
      for (int q = 0; q < 3; ++q) c->tick_prev_x0[q] = static_cast<double>(rx[q]);
    }
  }
  if (coords_out != nullptr)
    std::memcpy(coords_out, from_map ? reinterpret_cast<const void*>(c->h_tick_out + out.coords)
                                     : reinterpret_cast<const void*>(coords),
                static_cast<size_t>(H) * 3 * sizeof(double));
  return ACMPC_OK;
}

int acmpc_bind_map(acmpc_ctx* c, const double* centre, int32_t M, double spacing) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (centre == nullptr || M < 3 || !(spacing > 0.0)) return fail(c, ACMPC_EINVAL, "need a centre line of >= 3 points and a positive spacing");
  c->h_map.assign(centre, centre + 2 * static_cast<size_t>(M));
  c->map_spacing = spacing;
  c->map_dirty = true;
  return ACMPC_OK;
}

int acmpc_map_reference_path(acmpc_ctx* c, int32_t map_index, double pose_x, double pose_y, double lateral_offset,
                             int32_t horizon, int32_t centreline_points, double* coords, int32_t* first_index) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (coords == nullptr) return fail(c, ACMPC_EINVAL, "null output");
  const int n = horizon - 1;
  if (horizon < 3 || n > std::min(c->prm.max_steps, acmpc::kPrologueMaxSteps))
    return fail(c, ACMPC_ECAPACITY, "horizon out of range for this handle");
  int rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_staging(c);
  if (rc != ACMPC_OK) return rc;
  rc = ensure_tick(c);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = c->stream;
  const TickOutLayout out(n);
  acmpc::MapWindowArgs ma{};
  rc = map_window_args(c, horizon, centreline_points, out, &ma);
  if (rc != ACMPC_OK) return rc;
  rc = upload_map(c, s);
  if (rc != ACMPC_OK) return rc;
  ma.centre = c->d_map;
  acmpc::TickHeader* h = reinterpret_cast<acmpc::TickHeader*>(c->h_tick);
  h->use_map = 1;
  h->map_index = map_index;
  h->pose_x = pose_x;
  h->pose_y = pose_y;
  h->lateral_offset = lateral_offset;
  ACMPC_HIP(c, acmpc::launch_map_window(ma, s));
  ACMPC_HIP(c, hipStreamSynchronize(s));
  std::memcpy(coords, c->h_tick_out + out.coords, static_cast<size_t>(horizon) * 3 * sizeof(double));
  if (first_index != nullptr) *first_index = reinterpret_cast<const int*>(c->h_tick_out + out.status)[2];
  return ACMPC_OK;
}

int acmpc_tick_read_device_tables(acmpc_ctx* c, float* x0, float* u_ref, float* coef) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (x0 == nullptr || u_ref == nullptr || coef == nullptr) return fail(c, ACMPC_EINVAL, "null output");
  if (!c->tick_ready || c->tick_last_n == 0) return fail(c, ACMPC_ESTATE, "acmpc_control_tick has not run");
  const int n = c->tick_last_n;
  const TickLayout in(n, c->coef_stride);
  ACMPC_HIP(c, hipMemcpy(x0, c->d_tick + in.x0, 3 * sizeof(float), hipMemcpyDeviceToHost));
  ACMPC_HIP(c, hipMemcpy(u_ref, c->d_tick + in.uref, static_cast<size_t>(n) * 2 * sizeof(float), hipMemcpyDeviceToHost));
  ACMPC_HIP(c, hipMemcpy(coef, c->d_tick + in.coef, static_cast<size_t>(n) * c->coef_stride * sizeof(float),
                         hipMemcpyDeviceToHost));
  return ACMPC_OK;
}

int acmpc_tick_read_device_frames(acmpc_ctx* c, float* out, int64_t capacity_floats) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (out == nullptr) return fail(c, ACMPC_EINVAL, "null output");
  if (!c->tick_ready || c->tick_last_n == 0) return fail(c, ACMPC_ESTATE, "acmpc_control_tick has not run");
  const int n = c->tick_last_n;
  if (c->prm.mode != ACMPC_MODE_TEMPORAL || c->prm.nn_ahead >= 0 || n < acmpc::kVerifiedWindow || !acmpc::trio_frames_fit(n) ||
      c->opt.no_trio_rounds || c->sw.no_verified_search)
    return fail(c, ACMPC_ESTATE, "the last tick tabulated no frames (mode T with the exhaustive search, window <= n <= 106 steps)");
  const int floats = acmpc::verified_frame_floats(n);
  if (capacity_floats < floats) return fail(c, ACMPC_ECAPACITY, "output buffer too small");
  const TickLayout in(n, c->coef_stride);
  ACMPC_HIP(c, hipMemcpy(out, c->d_tick + in.frames, static_cast<size_t>(floats) * sizeof(float), hipMemcpyDeviceToHost));
  return ACMPC_OK;
}

int acmpc_speed_profile_qp_device(acmpc_ctx* c, const double* v_hi, const double* ds, int32_t n, double a_min,
                                  double a_max, double v_min, int32_t max_iter, int32_t check_every, double eps_abs,
                                  double eps_rel, double* v, double* y, int32_t warm_start, int32_t* iterations) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (v_hi == nullptr || ds == nullptr || v == nullptr || y == nullptr || n < 2) return fail(c, ACMPC_EINVAL, "bad argument");
  if (n > acmpc::kPrologueMaxSteps) return fail(c, ACMPC_ECAPACITY, "the device solver holds at most 128 points");
  int rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  double* d = nullptr;  // v_hi | ds | v | y | status
  const size_t doubles = static_cast<size_t>(5) * n + 2;
  ACMPC_HIP(c, hipMalloc(reinterpret_cast<void**>(&d), doubles * sizeof(double)));
  double *d_vhi = d, *d_ds = d + n, *d_v = d + 2 * n, *d_y = d + 3 * n;
  int* d_out = reinterpret_cast<int*>(d + 5 * n);
  hipError_t e = hipMemcpy(d_vhi, v_hi, static_cast<size_t>(n) * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ds, ds, static_cast<size_t>(n) * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_v, v, static_cast<size_t>(n) * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_y, y, static_cast<size_t>(2 * n - 1) * sizeof(double), hipMemcpyHostToDevice);
  const acmpc::admm::Settings st{a_min, a_max, v_min, max_iter, check_every > 0 ? check_every : 10, eps_abs, eps_rel};
  if (e == hipSuccess) e = acmpc::launch_admm(d_vhi, d_ds, n, st, d_v, d_y, warm_start, d_out, nullptr);
  int out[2] = {1, 0};
  if (e == hipSuccess) e = hipMemcpy(v, d_v, static_cast<size_t>(n) * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(y, d_y, static_cast<size_t>(2 * n - 1) * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail_hip(c, e, "acmpc_speed_profile_qp_device");
  if (iterations != nullptr) *iterations = out[1];
  return out[0];
}

extern "C++" {
namespace {

using AllReduceFn = ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
using ErrorStringFn = const char* (*)(ncclResult_t);

using UniqueIdFn = ncclResult_t (*)(ncclUniqueId*);
using CommInitRankFn = ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int);
using CommDestroyFn = ncclResult_t (*)(ncclComm_t);

struct Rccl {
  AllReduceFn all_reduce = nullptr;
  ErrorStringFn error_string = nullptr;
  UniqueIdFn unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
};

// the RCCL that is already in the process owns the caller's communicator; only without one open the system's
const Rccl& rccl() {
  static const Rccl api = [] {
    Rccl r;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    void* handle = nullptr;
    if (sym == nullptr) {
      const char* path = std::getenv("ACMPC_RCCL_LIBRARY");
      handle = dlopen(path != nullptr ? path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
      if (handle == nullptr && path == nullptr) handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
      if (handle != nullptr) sym = dlsym(handle, "ncclAllReduce");
    }
    r.all_reduce = reinterpret_cast<AllReduceFn>(sym);
    auto also = [handle](const char* name) { return (handle != nullptr) ? dlsym(handle, name) : dlsym(RTLD_DEFAULT, name); };
    r.error_string = reinterpret_cast<ErrorStringFn>(also("ncclGetErrorString"));
    r.unique_id = reinterpret_cast<UniqueIdFn>(also("ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(also("ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(also("ncclCommDestroy"));
    return r;
  }();
  return api;
}

}  // namespace
}  // extern "C++"

// A communicator for acmpc_reduce_across_ranks from the SAME copy of RCCL that call resolves (a process can hold two - the
// system's and the one PyTorch bundles - and a communicator only works with the copy that made it).
int acmpc_rccl_unique_id(void* id_out) {
  if (id_out == nullptr) return ACMPC_EINVAL;
  const Rccl& api = rccl();
  if (api.unique_id == nullptr) return ACMPC_ESTATE;
  static_assert(sizeof(ncclUniqueId) == ACMPC_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  return api.unique_id(static_cast<ncclUniqueId*>(id_out)) == ncclSuccess ? ACMPC_OK : ACMPC_EHIP;
}

int acmpc_rccl_comm_create(const void* id, int32_t n_ranks, int32_t rank, int32_t device, void** comm_out) {
  if (id == nullptr || comm_out == nullptr || n_ranks < 1 || rank < 0 || rank >= n_ranks) return ACMPC_EINVAL;
  *comm_out = nullptr;
  const Rccl& api = rccl();
  if (api.comm_init_rank == nullptr) return ACMPC_ESTATE;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return ACMPC_ENODEVICE;
  }
  ncclUniqueId by_value;
  std::memcpy(&by_value, id, sizeof by_value);
  ncclComm_t comm = nullptr;
  if (api.comm_init_rank(&comm, n_ranks, by_value, rank) != ncclSuccess) return ACMPC_EHIP;
  *comm_out = comm;
  return ACMPC_OK;
}

int acmpc_rccl_comm_destroy(void* comm) {
  if (comm == nullptr) return ACMPC_OK;
  const Rccl& api = rccl();
  if (api.comm_destroy == nullptr) return ACMPC_ESTATE;
  return api.comm_destroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? ACMPC_OK : ACMPC_EHIP;
}

int acmpc_reduce_across_ranks(acmpc_ctx* c, void* rccl_comm, int64_t* d_keys, int32_t P, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (rccl_comm == nullptr || d_keys == nullptr) return fail(c, ACMPC_EINVAL, "null communicator or keys");
  if (P < 1 || P > c->prm.max_problems) return fail(c, ACMPC_ECAPACITY, "P exceeds the handle's capacity");
  const Rccl& api = rccl();
  if (api.all_reduce == nullptr) return fail(c, ACMPC_ESTATE, "no RCCL in the process and librccl.so.1 not loadable");
  const ncclResult_t rc = api.all_reduce(d_keys, d_keys, static_cast<size_t>(P), ncclInt64, ncclMin,
                                         static_cast<ncclComm_t>(rccl_comm), static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) {
    std::string msg = "ncclAllReduce: ";
    msg += (api.error_string != nullptr) ? api.error_string(rc) : "error";
    return fail(c, ACMPC_EHIP, msg.c_str());
  }
  return ACMPC_OK;
}

int acmpc_rollout_start_clocks(acmpc_ctx* c, uint64_t* out, int32_t capacity, int32_t* count) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (out == nullptr || count == nullptr) return fail(c, ACMPC_EINVAL, "null output");
  *count = c->start_clock_count;
  if (c->start_clock_count == 0) return ACMPC_OK;
  if (capacity < c->start_clock_count) return fail(c, ACMPC_ECAPACITY, "start clocks: capacity below the launch's workgroups");
  ACMPC_HIP(c, hipDeviceSynchronize());
  ACMPC_HIP(c, hipMemcpy(out, c->d_start_clock, static_cast<size_t>(c->start_clock_count) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return ACMPC_OK;
}

int acmpc_profile_enable(acmpc_ctx* c, int32_t capacity) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (capacity < 0) return fail(c, ACMPC_EINVAL, "negative capacity");
  const int rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  while (static_cast<int32_t>(c->prof_start.size()) < capacity) {
    hipEvent_t e0, e1;
    ACMPC_HIP(c, hipEventCreate(&e0));
    ACMPC_HIP(c, hipEventCreate(&e1));
    c->prof_start.push_back(e0);
    c->prof_stop.push_back(e1);
  }
  while (static_cast<int32_t>(c->prof_start.size()) > capacity) {
    (void)hipEventDestroy(c->prof_start.back());
    (void)hipEventDestroy(c->prof_stop.back());
    c->prof_start.pop_back();
    c->prof_stop.pop_back();
  }
  c->prof_used = 0;
  return ACMPC_OK;
}

int acmpc_profile_collect(acmpc_ctx* c, float* out_ms, int32_t capacity, int32_t* count) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (out_ms == nullptr || count == nullptr) return fail(c, ACMPC_EINVAL, "null output");
  const int32_t used = static_cast<int32_t>(c->prof_used);
  const int32_t take = used < capacity ? used : capacity;
  for (int32_t i = 0; i < take; ++i) {
    ACMPC_HIP(c, hipEventSynchronize(c->prof_stop[i]));
    ACMPC_HIP(c, hipEventElapsedTime(&out_ms[i], c->prof_start[i], c->prof_stop[i]));
  }
  *count = take;
  c->prof_used = 0;
  return ACMPC_OK;
}

int acmpc_softmin_device(acmpc_ctx* c, const float* d_costs, const int64_t* d_keys, const float* d_U, int32_t P,
                         int32_t N, int32_t n, int32_t layout, float* d_mean, double* d_weight_sum, void* stream) {
  if (c == nullptr) return ACMPC_EINVAL;
  if (d_costs == nullptr || d_keys == nullptr || d_U == nullptr || d_mean == nullptr)
    return fail(c, ACMPC_EINVAL, "null device pointer");
  int rc = check_shape(c, P, N, n, layout);
  if (rc != ACMPC_OK) return rc;
  if (!(c->prm.softmin_lambda > 0.0)) return fail(c, ACMPC_EINVAL, "softmin_lambda must be positive");
  rc = ensure_device(c);
  if (rc != ACMPC_OK) return rc;
  acmpc::SoftminArgs a{};
  a.costs = d_costs;
  a.keys = d_keys;
  a.U = d_U;
  a.partial = c->d_soft_partial;
  a.mean = d_mean;
  a.weight_sum = d_weight_sum;
  a.chunks = acmpc::softmin_chunks(N);
  a.P = P;
  a.N = N;
  a.n = n;
  a.lambda = static_cast<float>(c->prm.softmin_lambda);
  ACMPC_HIP(c, acmpc::launch_softmin(layout, a, static_cast<hipStream_t>(stream)));
  return ACMPC_OK;
}

}  // extern "C"
