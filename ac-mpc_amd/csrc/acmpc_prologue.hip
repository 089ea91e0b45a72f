// Per-tick prologue of the closed-loop solve, on the device: everything
// SpatialMPC.get_control does between receiving the H x 3 reference path and calling the control solver
// (src/acmpc/control/spatial_mpc.py:180-191), as ONE single-wavefront kernel that is the first node of the solve's
// hipGraph:
//
//   construct_waypoints   spatial_mpc.py:125-154      H x 3 (x, y, width) -> 7 x n table
//   velocity ceiling      speed_profile.py:26-43,131-150
//   speed-profile QP      speed_profile.py:47-59      tridiagonal ADMM (acmpc_admm.h), warm-started from the state this
//                                                     handle keeps on the device between ticks
//   t2s                   dynamics.py:23-40           pose (offset, 0, pi/2) -> Frenet start state
//   linearise + corridor  dynamics.py:65-103, control.py:57-60   -> the packed float32 table the rollout reads
//   reference controls    control.py:26-33            u_ref = clip((v_ref, kappa), input box)
//
// float64 throughout, the same operations in the same order as the host path (acmpc_waypoint_table,
// acmpc_velocity_ceiling, acmpc_speed_profile_qp, acmpc_set_paths); results are written as float32 straight into the
// staging block the rollout kernels of the same graph read, and the 7 x n table goes to pinned host memory for the
// caller's read-after-call attributes.  The device's atan2 / sin / cos may differ from the host libm's in the last
// float64 bit; everything else (+, -, *, /, sqrt, fmod, comparisons) is IEEE-exact on both sides.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "acmpc_admm.h"
#include "acmpc_frames.h"
#include "acmpc_prologue.h"

#pragma clang fp contract(off)

namespace acmpc {

// Phase stamps of the prologue for tools/archive/prologue_probe.py (an A/B build with -DACMPC_STAMPS); nothing in the library.
#ifdef ACMPC_STAMPS
__device__ unsigned long long g_prologue_stamps[16];
#define ACMPC_PSTAMP(slot)                                                    \
  do {                                                                        \
    if (threadIdx.x == 0) g_prologue_stamps[(slot)] = wall_clock64();         \
  } while (0)
#else
#define ACMPC_PSTAMP(slot) \
  do {                     \
  } while (0)
#endif

namespace {

constexpr double kPi = 3.14159265358979323846;

// np.mod(a + pi, 2 pi) - pi: floored modulo, result in [-pi, pi)
__device__ __forceinline__ double wrap_angle(double a) {
  const double two_pi = 2.0 * kPi;
  double m = fmod(a + kPi, two_pi);
  if (m != 0.0 && (m < 0.0)) m += two_pi;
  return m - kPi;
}

struct WaveTeam {
  static constexpr int size = 64;
  __device__ __forceinline__ int rank() const { return static_cast<int>(threadIdx.x); }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ double max(double v, double*) const {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const double o = __shfl_xor(v, m, 64);
      v = o > v ? o : v;
    }
    return v;
  }
};

}  // namespace

// ---- mode T with the exhaustive nearest-waypoint search: the frames of its verified window search (acmpc_frames.h, the
// arithmetic acmpc_set_paths runs on the host), one or two windows per lane.  xr / yr: the n float32 waypoint positions the
// rollout sees, widened (LDS); gap_doubles: 4 n + 1 doubles of LDS scratch.  One wavefront: a workgroup of its own in the
// prologue's launch (the frames need the path, not the speed profile: they are ready before the first workgroup is) ----
__device__ void tabulate_frames(const double* xr, const double* yr, double* gap_doubles, int n, int lane, float* out) {
  const WaveTeam team;
  constexpr int W = kVerifiedWindow;
  const int windows = n - W + 1;
  const double inf = __builtin_huge_val();
  auto at = [&](int mm, double& x, double& y) {
    x = xr[mm];
    y = yr[mm];
  };
  double wn = 0.0;
  bool finite = true;
  for (int i = lane; i < n; i += 64) {
    const double x = xr[i], y = yr[i];
    finite = finite && (fabs(x) < inf) && (fabs(y) < inf);
    const double norm = sqrt(x * x + y * y);
    wn = (norm > wn) ? norm : wn;
  }
  wn = team.max(wn, nullptr);
  finite = __ballot(!finite) == 0ull;
  float largest_gap2 = 0.0f;   // (between neighbours: decides the first choice of `near`, acmpc_frames.h)
  for (int i = lane; i + 1 < n; i += 64) {
    const float gap2 = frames::squared_gap(static_cast<float>(xr[i + 1]), static_cast<float>(yr[i + 1]),
                                           static_cast<float>(xr[i]), static_cast<float>(yr[i]));
    largest_gap2 = (gap2 > largest_gap2) ? gap2 : largest_gap2;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float other = __shfl_xor(largest_gap2, m, 64);
    largest_gap2 = (other > largest_gap2) ? other : largest_gap2;
  }
  const int near0 = __builtin_amdgcn_readfirstlane(frames::near_first(largest_gap2));
  // The far split for near = near0 (the path's first choice), every window at once.  far_distance() walks (far waypoint, window waypoint)
  // pairs window by window - up to 200 per window, on one lane.  Turned round: waypoint q = lo + j of a window needs the
  // nearest waypoint m >= q + near + W - j ahead and m <= q - near - 1 - j behind, j = 0 .. W - 1; the lane that holds
  // waypoint q finds those 2 W minima in one sweep over the path (n pairs), and a window's R^2 is the minimum over its
  // W waypoints' entries.  The same pairs through the same squared_gap(): the same minimum, whatever the order.  The
  // sweep's index is wave-uniform, so waypoint m comes out of its lane's registers (v_readlane), not out of LDS: this is
  // one wavefront, and every LDS round trip it cannot overlap it waits for.
  constexpr int kSlots = (kPrologueMaxSteps + 63) / 64;   // waypoints (and windows) per lane
  const int slots = (n + 63) / 64;                         // (wave-uniform)
  float own_x[kSlots], own_y[kSlots];
#pragma unroll
  for (int r = 0; r < kSlots; ++r) {
    const int q = min(lane + 64 * r, n - 1);
    own_x[r] = static_cast<float>(xr[q]);
    own_y[r] = static_cast<float>(yr[q]);
  }
  auto lane_value = [](float v, int src) {   // v of lane `src` (wave-uniform)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  };
  auto waypoint = [&](int mm, float& x, float& y) {   // mm wave-uniform
    const int src = mm & 63;
    x = lane_value((mm < 64) ? own_x[0] : own_x[kSlots - 1], src);
    y = lane_value((mm < 64) ? own_y[0] : own_y[kSlots - 1], src);
  };
  static_assert(kSlots <= 2, "waypoint(): two slots per lane");
  float* gap = reinterpret_cast<float*>(gap_doubles);   // [n][W] minima of one sweep, + one slot nobody reads
  const int nowhere = n * W;
  const float finf = __builtin_inff();
  float r2[kSlots];
#pragma unroll
  for (int q = 0; q < kSlots; ++q) r2[q] = finf;
  // (no branch inside a sweep: a lone wavefront pays tens of cycles for each; what does not apply is stored `nowhere`)
  auto sweep = [&](auto slots_tag, auto pass_tag) {
    constexpr int kLive = decltype(slots_tag)::value;   // slots in use: 1 up to 64 waypoints, 2 beyond
    constexpr bool kAhead = decltype(pass_tag)::value == 0;
    float running[kLive];
#pragma unroll
    for (int r = 0; r < kLive; ++r) running[r] = finf;
    const int count = n - 1 - near0;
    for (int it = 0, mm = kAhead ? n - 1 : 0; it < count; ++it, mm += kAhead ? -1 : 1) {
      float xm, ym;
      waypoint(mm, xm, ym);
#pragma unroll
      for (int r = 0; r < kLive; ++r) {
        const int q = lane + 64 * r;
        // ahead: m > q + near, entry j = q + near + W - m (< W);  behind: m < q - near, entry j = q - near - 1 - m (>= 0)
        const int j = kAhead ? q + near0 + W - mm : q - near0 - 1 - mm;
        const int applies = static_cast<int>(q < n) & static_cast<int>(kAhead ? mm > q + near0
                                                                              : mm < q - near0);
        const float d2 = frames::squared_gap(xm, ym, own_x[r], own_y[r]);
        running[r] = ((applies & static_cast<int>(d2 < running[r])) != 0) ? d2 : running[r];
        const int kept = applies & static_cast<int>(j >= 0) & static_cast<int>(j < W);
        gap[(kept != 0) ? q * W + j : nowhere] = running[r];
      }
    }
  };
  auto sweep_either = [&](auto pass_tag) {
    if (slots == 1) {
      sweep(std::integral_constant<int, 1>{}, pass_tag);
    } else {
      sweep(std::integral_constant<int, kSlots>{}, pass_tag);
    }
  };
  for (int pass = 0; pass < 2; ++pass) {   // 0: the waypoints ahead, 1: those behind (one [n][W] buffer for both)
    for (int e = lane; e < n * W; e += 64) gap[e] = finf;
    team.sync();
    if (pass == 0) {
      sweep_either(std::integral_constant<int, 0>{});
    } else {
      sweep_either(std::integral_constant<int, 1>{});
    }
    team.sync();
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      const int lo = lane + 64 * q;
      if (lo < windows) {
        float g[W];
#pragma unroll
        for (int j = 0; j < W; ++j) g[j] = gap[(lo + j) * W + j];
#pragma unroll
        for (int j = 0; j < W; ++j) r2[q] = (g[j] < r2[q]) ? g[j] : r2[q];
      }
    }
    team.sync();
  }
  frames::Geometry geometry[kSlots];
  double slab_max = 0.0;
#pragma unroll
  for (int q = 0; q < kSlots; ++q) {
    const int lo = lane + 64 * q;
    geometry[q].usable = false;
    const bool mine = lo < windows;
    double R = (mine && finite) ? frames::far_from_squared(r2[q]) : inf;
    const int near = (mine && finite) ? frames::choose_near(at, n, lo, near0, R) : near0;
    // (the usual cases - every window's first choice of `near` stands, 16 or 32 waypoints - take the unrolled forms)
    const bool first_choice_stands = __ballot(mine && near != near0) == 0ull;
    if (first_choice_stands && near0 == 16) {
      if (mine) geometry[q] = frames::window_geometry<16>(at, n, lo, finite, near, R);
    } else if (first_choice_stands && near0 == 32) {
      if (mine) geometry[q] = frames::window_geometry<32>(at, n, lo, finite, near, R);
    } else if (mine) {
      geometry[q] = frames::window_geometry<0>(at, n, lo, finite, near, R);
    }
    const double slab = geometry[q].aA - geometry[q].aB;
    if (mine && geometry[q].usable) slab_max = (slab > slab_max) ? slab : slab_max;
  }
  slab_max = team.max(slab_max, nullptr);
  const frames::Scale scale = frames::path_scale(wn, slab_max);
#pragma unroll
  for (int q = 0; q < kSlots; ++q) {
    const int lo = lane + 64 * q;
    if (lo < windows) frames::frame_row(geometry[q], scale, out + static_cast<size_t>(kFrameStride) * lo);
  }
}

__global__ void __launch_bounds__(64) prologue_kernel(const PrologueArgs a) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  // ONE read of the head (kernel arguments, or the pinned block over the host link); everything below uses the copy
  const TickHeader h = (a.header_by_value != 0) ? a.header_value : *a.header;
  ACMPC_PSTAMP(0);
  const int lane = static_cast<int>(threadIdx.x);
  const int H = __builtin_amdgcn_readfirstlane(h.horizon);   // (wave-uniform: loop bounds on the scalar unit)
  const int n = H - 1;
  const int m = n - 1;
  if (blockIdx.x == 1) {
    // ---- the second workgroup (launched for mode T with the exhaustive search): the search frames, from the path alone ----
    if (a.frames == nullptr || n < kVerifiedWindow) return;
    double* xr = s_mem;
    double* yr = s_mem + n;
    if (h.use_map != 0) {   // the rows of the map window, as the first workgroup computes them
      int first = (h.map_index < 0) ? a.map_first[0] : h.map_index;
      first = ((first % a.map_M) + a.map_M) % a.map_M;
      const MapFrame frame = map_frame(a.map_centre, a.map_M, first);
      for (int r = lane; r < n; r += 64) {
        double row[3];
        map_path_row(a.map_centre, a.map_M, first, a.map_count, a.map_points, H, r, h.lateral_offset, frame, row);
        xr[r] = static_cast<double>(static_cast<float>(row[0]));
        yr[r] = static_cast<double>(static_cast<float>(row[1]));
      }
    } else {
      for (int r = lane; r < n; r += 64) {   // construct_waypoints: x, y = the first n points of the path
        xr[r] = static_cast<double>(static_cast<float>((a.path_by_value != 0) ? a.coords_value[3 * r] : a.coords[3 * r]));
        yr[r] = static_cast<double>(static_cast<float>((a.path_by_value != 0) ? a.coords_value[3 * r + 1] : a.coords[3 * r + 1]));
      }
    }
    __syncthreads();
    {   // the path's own frame, as the kernels take it (acmpc_device.h: start_temporal): float32 differences to waypoint 0
      const float ox = static_cast<float>(xr[0]), oy = static_cast<float>(yr[0]);
      __syncthreads();
      for (int r = lane; r < n; r += 64) {
        xr[r] = static_cast<double>(static_cast<float>(xr[r]) - ox);
        yr[r] = static_cast<double>(static_cast<float>(yr[r]) - oy);
      }
      __syncthreads();
    }
    tabulate_frames(xr, yr, s_mem + 2 * n, n, lane, a.frames);
    return;
  }
  // LDS: table [7][n] | v_hi [n] | psi_seg [n + 1] (heading of every segment incl. the closing one) | v [n] |
  //      y [2n - 1] | ADMM workspace
  double* table = s_mem;
  double *tx = table, *ty = table + n, *tpsi = table + 2 * n, *tkappa = table + 3 * n, *tds = table + 4 * n,
         *twidth = table + 5 * n, *tv = table + 6 * n;
  double* v_hi = table + 7 * n;
  double* seg = v_hi + n;
  double* qv = seg + (n + 1);
  double* qy = qv + n;
  admm::Workspace ws;
  ws.bind(qy + 2 * n, n);
  double* s_coords = qy + 2 * n + admm::workspace_doubles(n);   // [H][3]: the path when it comes from the map
  const double* __restrict__ coords = a.coords;
  const WaveTeam team;
  // Requested now, used at the end: the previous plan (pinned host memory: a trip over the host link) and the previous
  // iterate of the speed-profile solver (device memory).  Read where they are needed, each was a wait of its own on the
  // one wavefront's critical path; this way they arrive while the waypoints are built.  (Lane i holds entries i, i + 64.)
  constexpr int kPerLane = (kPrologueMaxSteps + 63) / 64;
  float centre_in_v[kPerLane], centre_in_k[kPerLane];
  double warm_v[kPerLane], warm_y[2 * kPerLane];
  double* state = a.warm_state + static_cast<size_t>(h.localised != 0 ? 1 : 0) * a.warm_stride;
  const double warm_flag = state[0], warm_n = state[1];
#pragma unroll
  for (int q = 0; q < kPerLane; ++q) {
    const int i = min(lane + 64 * q, n - 1);
    if (a.path_by_value != 0) {   // (i < kInlinePathPoints: the host checked the horizon)
      centre_in_v[q] = (h.centre_is_reference != 0) ? 0.0f : a.centre_value[2 * min(i, kInlinePathPoints - 1)];
      centre_in_k[q] = (h.centre_is_reference != 0) ? 0.0f : a.centre_value[2 * min(i, kInlinePathPoints - 1) + 1];
    } else {
      centre_in_v[q] = (h.centre_is_reference != 0) ? 0.0f : a.centre_in[2 * i];
      centre_in_k[q] = (h.centre_is_reference != 0) ? 0.0f : a.centre_in[2 * i + 1];
    }
    warm_v[q] = (a.warm_capacity >= n) ? state[2 + i] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < 2 * kPerLane; ++q)
    warm_y[q] = (a.warm_capacity >= n) ? state[2 + n + min(lane + 64 * q, 2 * n - 2)] : 0.0;
  if (h.use_map != 0) {
    // the reference path cut out of the bound map by this wavefront itself (no launch in front of the prologue)
    int first = (h.map_index < 0) ? a.map_first[0] : h.map_index;
    first = ((first % a.map_M) + a.map_M) % a.map_M;
    const MapFrame frame = map_frame(a.map_centre, a.map_M, first);
    for (int r = lane; r < H; r += 64) {
      double row[3];
      map_path_row(a.map_centre, a.map_M, first, a.map_count, a.map_points, H, r, h.lateral_offset, frame, row);
      for (int e = 0; e < 3; ++e) {
        s_coords[3 * r + e] = row[e];
        a.coords_out[3 * r + e] = row[e];
      }
    }
    if (lane == 0) a.index_out[0] = first;
    team.sync();
    coords = s_coords;
  } else if (a.path_by_value != 0) {
    // the caller's path out of the kernel arguments (device memory) instead of the pinned block (the host link)
    for (int r = lane; r < 3 * H; r += 64) s_coords[r] = a.coords_value[r];
    team.sync();
    coords = s_coords;
  }

  ACMPC_PSTAMP(1);
  // ---- construct_waypoints (spatial_mpc.py:125-154) ------------------------------------------------------------
  // headings of the n forward segments and of the segment that closes the loop from the last point to point 0
  for (int i = lane; i <= n; i += 64) {
    const double* here = coords + 3 * ((i == n) ? 0 : i);
    const double* from = coords + 3 * ((i == n) ? H - 1 : i);
    const double* to = coords + 3 * ((i == n) ? 0 : i + 1);
    const double ax = to[0] - from[0], ay = to[1] - from[1];
    seg[i] = atan2(ay, ax);
    if (i < n) {
      tx[i] = here[0];
      ty[i] = here[1];
      tds[i] = sqrt(ax * ax + ay * ay);
      twidth[i] = to[2];
    }
  }
  team.sync();
  for (int i = lane; i < n; i += 64) {
    const double psi = seg[i];
    const double behind = seg[(i == 0) ? n : i - 1];  // atan2 of (here - prev): the forward segment of point i - 1
    tpsi[i] = psi;
    tkappa[i] = wrap_angle(psi - behind) / (tds[i] + h.eps) + h.eps;
  }
  team.sync();
  if (lane == 0) tkappa[0] = tkappa[1];
  team.sync();

  ACMPC_PSTAMP(2);
  // ---- velocity ceiling (speed_profile.py:26-43, localised: 131-150) ----------------------------------------------
  for (int i = lane; i < n; i += 64) {
    double c;
    if (h.localised != 0) {
      c = h.v_max;
    } else {
      const double curvature = fabs(tkappa[i]);
      c = sqrt(h.ay_max / (curvature + 1e-12));
      if (curvature < h.ki_min) c = h.v_max;
      c = fmin(c, h.v_max);
      c = fmax(h.v_min, c);
      c = c + 2.0;
      if (h.has_end_velocity != 0 && i == n - 1) c = h.end_velocity;
    }
    v_hi[i] = c;
  }
  // ---- speed-profile QP, warm-started from the previous tick's iterate of the same solver ----------------------------
  // state: [valid, n] [v n] [y 2n - 1]
  const bool warm = warm_flag == 1.0 && warm_n == static_cast<double>(n) && a.warm_capacity >= n;
  if (warm) {
#pragma unroll
    for (int q = 0; q < kPerLane; ++q)
      if (lane + 64 * q < n) qv[lane + 64 * q] = warm_v[q];
#pragma unroll
    for (int q = 0; q < 2 * kPerLane; ++q)
      if (lane + 64 * q < 2 * n - 1) qy[lane + 64 * q] = warm_y[q];
  }
  team.sync();
  ACMPC_PSTAMP(3);
  const admm::Settings settings{h.a_min, h.a_max, h.v_min, h.qp_max_iter, h.qp_check_every > 0 ? h.qp_check_every : 10,
                                h.qp_eps_abs, h.qp_eps_rel};
  int iterations = 0;
  // the QP's exact optimum in two sweeps where it has that shape (every tick of a feasible profile: no iteration count to
  // depend on the approach of a braking zone); the splitting - with the iterate kept from the last tick it ran - otherwise
  const bool swept = h.qp_method == 0 && admm::exact_profile(team, ws, v_hi, tds, n, settings, qv, qy, ws.ex[0]);
  const int status = swept ? 0 : admm::solve(team, ws, v_hi, tds, n, settings, qv, qy, warm ? 1 : 0, &iterations);
  ACMPC_PSTAMP(4);
  if (status == 0 && a.warm_capacity >= n) {  // keep the iterate only when solved, as the host solver object does
    for (int i = lane; i < n; i += 64) state[2 + i] = qv[i];
    for (int i = lane; i < 2 * n - 1; i += 64) state[2 + n + i] = qy[i];
    if (lane == 0) {
      state[0] = 1.0;
      state[1] = static_cast<double>(n);
    }
  }
  // an unsolved profile leaves the velocities the path was built with: zero (spatial_mpc.py:119-122,140)
  for (int i = lane; i < n; i += 64) tv[i] = (status == 0) ? qv[i] : 0.0;
  team.sync();

  ACMPC_PSTAMP(5);
  // ---- t2s of the pose (offset, 0, pi/2) w.r.t. waypoint 0 (dynamics.py:23-40, spatial_mpc.py:187) -------------------
  if (lane == 0) {
    const double wx = tx[0], wy = ty[0], wpsi = tpsi[0];
    const double lateral = cos(wpsi) * (0.0 - wy) - sin(wpsi) * (h.offset - wx);
    if (a.temporal != 0) {   // mode T rolls the pose itself: (X, Y, phi) = (offset, 0, pi / 2)  (spatial_mpc.py:185)
      a.x0[0] = static_cast<float>(h.offset);
      a.x0[1] = 0.0f;
      a.x0[2] = static_cast<float>(kPi / 2.0);
    } else {
      a.x0[0] = static_cast<float>(lateral);
      a.x0[1] = static_cast<float>(wrap_angle(kPi / 2.0 - wpsi));
      a.x0[2] = 0.0f;
    }
    a.status[0] = status;
    a.status[1] = iterations;
    a.seed[0] = h.seed_lo;
    a.seed[1] = h.seed_hi;
  }
  // ---- linearise + corridor rows + reference controls, rounded once to float32 (as acmpc_set_paths does) --------------
#pragma unroll
  for (int q = 0; q < kPerLane; ++q) {
    const int i = lane + 64 * q;
    if (i >= n) break;
    const double v = tv[i], ds = tds[i], kappa = tkappa[i], width = twidth[i];
    const double vds = v * ds + 1e-12;
    if (a.temporal != 0) {   // the waypoint rows of mode T (acmpc_set_paths' packing)
      float* out = a.coef + static_cast<size_t>(i) * 8;
      const double psi = tpsi[i];
      out[0] = static_cast<float>(tx[i]);
      out[1] = static_cast<float>(ty[i]);
      out[2] = static_cast<float>(cos(psi));
      out[3] = static_cast<float>(sin(psi));
      out[4] = static_cast<float>(psi);
      out[5] = static_cast<float>(kappa);
      out[6] = static_cast<float>(v);
      out[7] = static_cast<float>(width / 2.0 - a.margin);
    } else {
      float* out = a.coef + static_cast<size_t>(i) * 12;
      out[0] = static_cast<float>(ds);
      out[1] = static_cast<float>(-(kappa * kappa) * ds);
      out[2] = static_cast<float>(-kappa / vds);
      out[3] = static_cast<float>(-1.0 / (v * v * ds + 1e-12));
      out[4] = static_cast<float>(1.0 / vds);
      out[5] = static_cast<float>(v);
      out[6] = static_cast<float>(kappa);
      out[7] = static_cast<float>(-width / 2.0 + a.margin);
      out[8] = static_cast<float>(width / 2.0 - a.margin);
      out[9] = 0.0f;
      out[10] = 0.0f;
      out[11] = 0.0f;
    }
    const double uv = fmin(fmax(v, a.u_lo0), a.u_hi0);       // np.clip(velocities, lo, hi)
    const double uk = fmin(fmax(kappa, a.u_lo1), a.u_hi1);
    a.u_ref[2 * i] = static_cast<float>(uv);
    a.u_ref[2 * i + 1] = static_cast<float>(uk);
    // a solve without a previous plan samples round the reference controls
    a.centre[2 * i] = (h.centre_is_reference != 0) ? static_cast<float>(uv) : centre_in_v[q];
    a.centre[2 * i + 1] = (h.centre_is_reference != 0) ? static_cast<float>(uk) : centre_in_k[q];
  }
  ACMPC_PSTAMP(6);
  // the 7 x n table for the caller (pinned host memory: posted writes, visible once the stream has drained)
  for (int e = lane; e < 7 * n; e += 64) a.table_out[e] = table[e];
  ACMPC_PSTAMP(7);
  (void)m;
}

// ---- reference path from the map ----------------------------------------------------------------------
// What the perception stack and ControlProcess._reference_path do between the map and get_control, for a pose on a
// known map: the map point nearest to the pose (first minimum, what the localiser's KD-tree query returns,
// localiser.py:282-289), the next 150 m of centre line (perception/tracks.py:14) moved into the vehicle frame (car at the
// origin, heading +y), resampled to the 500 points perception publishes as float32 (controller.py:102-108), every
// (500 / H)-th of them kept and given widths linspace(10, 6, H) (controller.py:256-267).  One workgroup; row r of the
// path only needs the two window points that bracket its sample, so nothing but the nearest-point search is more than
// a handful of operations per lane.  float64 like the NumPy statement (workloads.local_centreline), rounded to float32
// where perception's shared memory does.
__global__ void __launch_bounds__(256) map_window_kernel(const MapWindowArgs a) {
  __shared__ double s_d[4];
  __shared__ int s_i[4];
  __shared__ int s_first;
  const int tid = static_cast<int>(threadIdx.x);
  const TickHeader h = *a.header;
  int first = h.map_index;
  if (first < 0) {  // nearest map point: per-thread scan, wave shuffles, four partial results through LDS
    double best = __builtin_inf();
    int best_i = 0x7fffffff;
    for (int m = tid; m < a.M; m += 256) {
      const double dx = h.pose_x - a.centre[2 * m], dy = h.pose_y - a.centre[2 * m + 1];
      const double d = dx * dx + dy * dy;
      if (d < best) {  // ascending m per thread: the first minimum stays
        best = d;
        best_i = m;
      }
    }
#pragma unroll
    for (int mask = 32; mask >= 1; mask >>= 1) {
      const double od = __shfl_xor(best, mask, 64);
      const int oi = __shfl_xor(best_i, mask, 64);
      if (od < best || (od == best && oi < best_i)) {
        best = od;
        best_i = oi;
      }
    }
    if ((tid & 63) == 0) {
      s_d[tid >> 6] = best;
      s_i[tid >> 6] = best_i;
    }
    __syncthreads();
    if (tid == 0) {
      for (int q = 1; q < 4; ++q)
        if (s_d[q] < best || (s_d[q] == best && s_i[q] < best_i)) {
          best = s_d[q];
          best_i = s_i[q];
        }
      s_first = best_i;
    }
    __syncthreads();
    first = s_first;
  }
  first = ((first % a.M) + a.M) % a.M;
  const MapFrame frame = map_frame(a.centre, a.M, first);
  for (int r = tid; r < a.H; r += 256) {
    double row[3];
    map_path_row(a.centre, a.M, first, a.count, a.points, a.H, r, h.lateral_offset, frame, row);
    for (int e = 0; e < 3; ++e) {
      if (a.coords != nullptr) a.coords[3 * r + e] = row[e];
      if (a.coords_out != nullptr) a.coords_out[3 * r + e] = row[e];
    }
  }
  if (tid == 0 && a.first_out != nullptr) a.first_out[0] = first;
}

#ifdef ACMPC_STAMPS
extern "C" int acmpc_debug_prologue_stamps(unsigned long long* out) {
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prologue_stamps), sizeof(unsigned long long) * 16));
}
#endif

hipError_t launch_map_window(const MapWindowArgs& args, hipStream_t s) {
  (void)hipGetLastError();
  hipLaunchKernelGGL(map_window_kernel, dim3(1), dim3(256), 0, s, args);
  return hipGetLastError();
}

size_t prologue_lds_bytes(int n) {
  return static_cast<size_t>(7 * n + n + (n + 1) + n + 2 * n + admm::workspace_doubles(n) + 3 * (n + 1)) * sizeof(double);  // < 64 kB
}

hipError_t launch_prologue(const PrologueArgs& args, int n, hipStream_t s) {
  (void)hipGetLastError();
  // (a second workgroup tabulates the search frames of mode T's exhaustive search while the first solves the speed profile)
  hipLaunchKernelGGL(prologue_kernel, dim3(args.frames != nullptr ? 2 : 1), dim3(64), prologue_lds_bytes(n), s, args);
  return hipGetLastError();
}

// ---- test hook: the device ADMM alone on caller-supplied (v_hi, ds) -----------------------------------------------------
__global__ void __launch_bounds__(64) admm_kernel(const double* v_hi_in, const double* ds_in, int n, admm::Settings s,
                                                  double* v, double* y, int warm, int* out) {
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  double* v_hi = s_mem;
  double* ds = v_hi + n;
  double* qv = ds + n;
  double* qy = qv + n;
  admm::Workspace ws;
  ws.bind(qy + 2 * n, n);
  const int lane = static_cast<int>(threadIdx.x);
  for (int i = lane; i < n; i += 64) {
    v_hi[i] = v_hi_in[i];
    ds[i] = ds_in[i];
    qv[i] = v[i];
  }
  for (int i = lane; i < 2 * n - 1; i += 64) qy[i] = y[i];
  __syncthreads();
  int iterations = 0;
  const int status = admm::solve(WaveTeam{}, ws, v_hi, ds, n, s, qv, qy, warm, &iterations);
  for (int i = lane; i < n; i += 64) v[i] = qv[i];
  for (int i = lane; i < 2 * n - 1; i += 64) y[i] = qy[i];
  if (lane == 0) {
    out[0] = status;
    out[1] = iterations;
  }
}

hipError_t launch_admm(const double* d_v_hi, const double* d_ds, int n, const admm::Settings& s, double* d_v, double* d_y,
                       int warm, int* d_out, hipStream_t stream) {
  (void)hipGetLastError();
  const size_t lds = static_cast<size_t>(5 * n + admm::workspace_doubles(n)) * sizeof(double);
  hipLaunchKernelGGL(admm_kernel, dim3(1), dim3(64), lds, stream, d_v_hi, d_ds, n, s, d_v, d_y, warm, d_out);
  return hipGetLastError();
}

}  // namespace acmpc
