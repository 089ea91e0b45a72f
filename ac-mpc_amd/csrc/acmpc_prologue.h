// Host-callable launcher of the per-tick prologue kernel (definitions in acmpc_prologue.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acmpc_admm.h"

namespace acmpc {

// Head of the per-tick input block in pinned host memory: everything that changes from tick to tick and therefore
// cannot be a kernel argument of the captured graph.  Followed in the block by the H x 3 reference path (float64) and
// the centre sequence [n][2] (float32).  The prologue kernel reads the block in place, over the host link (1.7 kB at
// H = 50: one round of loads) - no copy node in front of it - and leaves on the device what the rollout kernels read.
struct TickHeader {
  double offset;                          // lateral displacement of the car (spatial_mpc.py:187)
  double v_min, v_max, a_min, a_max, ay_max, ki_min, end_velocity;  // speed_profile_constraints (live dict)
  double qp_eps_abs, qp_eps_rel;
  double eps;                             // 1e-12 of construct_waypoints (spatial_mpc.py:34)
  int32_t horizon;                        // H; n = H - 1
  int32_t localised;                      // LocalisedSpeedProfileSolver instead of SpeedProfileSolver
  int32_t has_end_velocity;
  int32_t centre_is_reference;            // sample round the reference controls (no previous plan)
  int32_t qp_max_iter, qp_check_every;
  uint32_t seed_lo, seed_hi;              // Philox key of this solve (read by the rollout kernels through seed_ptr)
  int32_t qp_method;                      // 0: the speed profile's exact optimum in two sweeps first (acmpc_admm.h); 1: always the splitting
  int32_t reserved;
  // reference path taken from the bound map instead of from `coords` (acmpc_bind_map)
  int32_t use_map;                        // != 0: the window kernel in front of the prologue produces the H x 3 path
  int32_t map_index;                      // first waypoint of the window, or < 0: the map point nearest to the pose
  double pose_x, pose_y;                  // map frame; only read when map_index < 0
  double lateral_offset;                  // subtracted from the window's lateral coordinate (car off the centre line)
};

// The window kernel's arguments: map polyline resident on the device, output = the H x 3 path the prologue reads.
struct MapWindowArgs {
  const TickHeader* header;   // pinned host memory (use_map fields)
  const double* centre;       // [M][2] map centre line, device
  int M;
  int count;                  // map points in the look-ahead window: round(150 m / spacing) + 1 (perception/tracks.py:14)
  int points;                 // length of the resampled centre line perception publishes (500; controller.py:102-108)
  int H;                      // rows of the reference path: points / H must be an integer stride (controller.py:256-267)
  double* coords;             // [H][3] out, device (or nullptr)
  double* coords_out;         // [H][3] out, pinned host memory (or nullptr)
  int* first_out;             // [1] out: the window's first map index (device or pinned host memory; or nullptr)
};

struct PrologueArgs {
  // the head of the tick block: read in place from pinned host memory - or, when the kernel is launched directly (not
  // replayed from a captured graph), carried in the kernel arguments themselves, which saves the read over the host link
  int header_by_value;
  TickHeader header_value;
  // likewise the path and the previous plan of a directly launched tick, up to kInlinePathPoints points: the kernel
  // arguments live in device memory, the pinned block is a trip over the host link in front of construct_waypoints
  int path_by_value;
  double coords_value[3 * 64];   // [H][3], H <= kInlinePathPoints
  float centre_value[2 * 64];    // [n][2]
  const TickHeader* header;   // pinned host memory
  const double* coords;       // [H][3] (x, y, width), pinned host memory
  const float* centre_in;     // [n][2] pinned host memory (ignored when header->centre_is_reference)
  int temporal;               // != 0: a mode T handle - x0 is the pose itself and the table the [n][8] waypoint rows
  float* x0;                  // [3]        out: Frenet start state (mode T: the pose (offset, 0, pi / 2))
  float* u_ref;               // [n][2]     out: reference controls clipped to the input box
  float* coef;                // [n][12]    out: packed mode-S table (mode T: [n][8] = x, y, cos psi, sin psi, psi,
                              //            kappa, v, width / 2 - margin, as acmpc_set_paths packs it)
  float* frames;              // mode T, exhaustive search: [verified_frame_floats(n)] out, frames of the verified window
                              //            search (acmpc_frames.h); nullptr otherwise
  float* centre;              // [n][2]     out: the sequence round 0 samples round (centre_in or u_ref)
  uint32_t* seed;             // [2]        out: Philox key of this solve (the rollout kernels' seed_ptr)
  double* table_out;          // [7][n]     out, pinned host memory
  int* status;                // [2]        out, pinned host memory: QP status (0 solved), iterations
  double* warm_state;         // 2 solver slots x warm_stride doubles: [valid, n, v (n), y (2n - 1)]
  int warm_stride;
  int warm_capacity;          // largest n a slot can hold
  double margin;              // vehicle width / 2 (dynamics.py:14)
  double u_lo0, u_lo1, u_hi0, u_hi1;  // QP input box incl. the 0.1 m/s slack (control.py:130-139)
  // header.use_map: the H x 3 path is cut out of the map by this kernel's own lanes (see map_window_kernel)
  const double* map_centre;   // [M][2] device
  int map_M, map_count, map_points;
  const int* map_first;       // device: the window's first index when header.map_index < 0 (written by map_window_kernel)
  double* coords_out;         // [H][3] pinned host memory: the path used (map mode)
  int* index_out;             // pinned host memory
};

// Row r of the H x 3 reference path for the window of `count` map points that starts at map point `first`: the window moved
// into the vehicle frame (car at the window's first point heading +y), resampled to `points` samples the way np.interp does,
// every (points / H)-th kept, rounded to float32 where perception's shared memory does, widths linspace(10, 6, H)
// (workloads.local_centreline + ControlProcess._reference_path, controller.py:256-267).  ONE statement for the device - the
// window kernel's and the prologue's lanes - and the host, which cuts the same window for the plan of the last round's
// candidate 2 while the device builds the tick's tables (round 5; the two libm's atan2 / cos / sin may differ in the last
// float64 bit, which the float32 rounding of the positions absorbs but for a tie).
struct MapFrame {
  double x0, y0, c, sn;
};
__host__ __device__ inline MapFrame map_frame(const double* centre, int M, int first) {
  const double* w0 = centre + 2 * first;
  const double* w1 = centre + 2 * ((first + 1) % M);
  const double heading = atan2(w1[1] - w0[1], w1[0] - w0[0]);
  const double rot = 3.14159265358979323846 / 2.0 - heading;
  return MapFrame{w0[0], w0[1], cos(rot), sin(rot)};
}
__host__ __device__ inline void map_path_row(const double* centre, int M, int first, int count, int points, int H, int r,
                                             double lateral_offset, const MapFrame& f, double (&row)[3]) {
  const int stride = points / H;
  const double step = static_cast<double>(count - 1) / static_cast<double>(points - 1);  // np.linspace's step
  const int q = r * stride;                                   // sample of the resampled centre line kept for row r
  const double t = (q == points - 1) ? static_cast<double>(count - 1) : static_cast<double>(q) * step;
  int j = static_cast<int>(t);                                // np.interp: the bracket [j, j + 1] with xp = arange
  if (j > count - 2) j = count - 2;
  double local[2][2];
  for (int e = 0; e < 2; ++e) {
    const double* wp = centre + 2 * ((first + j + e) % M);
    const double dx = wp[0] - f.x0, dy = wp[1] - f.y0;
    local[e][0] = (dx * f.c + dy * (-f.sn)) - lateral_offset;   // (window - window[0]) @ [[c, s], [-s, c]]
    local[e][1] = dx * f.sn + dy * f.c;
  }
  const double frac = t - static_cast<double>(j);
  const bool last = t >= static_cast<double>(count - 1);
  const double x = last ? local[1][0] : (local[1][0] - local[0][0]) * frac + local[0][0];
  const double y = last ? local[1][1] : (local[1][1] - local[0][1]) * frac + local[0][1];
  row[0] = static_cast<double>(static_cast<float>(x));         // perception publishes float32
  row[1] = static_cast<double>(static_cast<float>(y));
  row[2] = (r == H - 1) ? 6.0 : 10.0 + static_cast<double>(r) * ((6.0 - 10.0) / static_cast<double>(H - 1));
}

constexpr int kInlinePathPoints = 64;
constexpr int kPrologueMaxSteps = admm::kPcrMaxN;  // LDS budget of the single-workgroup prologue (~390 n bytes) and the
                                                   // size up to which the tridiagonal solve is the parallel one

size_t prologue_lds_bytes(int n);
hipError_t launch_prologue(const PrologueArgs& args, int n, hipStream_t s);
hipError_t launch_map_window(const MapWindowArgs& args, hipStream_t s);

// test hook: the device ADMM alone (device pointers; v / y are read when warm != 0 and always written)
hipError_t launch_admm(const double* d_v_hi, const double* d_ds, int n, const admm::Settings& s, double* d_v, double* d_y,
                       int warm, int* d_out, hipStream_t stream);

}  // namespace acmpc
