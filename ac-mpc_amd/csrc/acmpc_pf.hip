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

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/acmpc.h"
#include "acmpc_device.h"  // Philox4x32-10, uniform_open, box_muller: the sampler's generator

namespace {

constexpr int kBlock = 256;
constexpr double kPi = 3.14159265358979323846;

struct Track {
  const double* xy;  // [m][2]
  int m;
};

// the map's points binned into square cells (see grid_nearest below)
struct GridIndex {
  const int* start;     // [nx * ny + 1]: cell c holds entries [start[c], start[c + 1]) of the three arrays below
  const double* x;      // the points in cell order (row-major cells, ascending point index inside a cell) ...
  const double* y;
  const int* index;     // ... and the index each has in the polyline
};
struct GridGeometry {
  double x0, y0, cell;
  int nx, ny;
};
// ... and, for the first ring of a search, the points of the 3 x 3 block of cells round every cell as ONE run (each point is
// in nine blocks: 6 MB for the 35 k-point map), all three polylines in one set of arrays: a query reads two bounds and one
// run instead of six bounds and three runs, and needs no choice of pointers by polyline
struct GridBlocks {
  const int* start;     // [3 * cells + 1]: block (t, c) holds entries [start[t * cells + c], start[t * cells + c + 1])
  const double* x;      // rows of the block in order, cells of a row in order, ascending point index inside a cell
  const double* y;
  const int* index;
  int cells;            // nx * ny
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
  const int* live;         // when not null: the number of particles is read from the device (resident filter)
  // nearest points found by pf_nearest_kernel (the grid search) - the scoring kernel then skips its own scan
  const int32_t* given_index;   // [P][3] or nullptr
  const double* given_d2;       // [P][3]
  // or found by the scoring workgroup itself through the grid (pf_score_kernel<1>: a wavefront per polyline)
  int own_grid_search;
  GridIndex grid[3];
  GridGeometry geometry;
  GridBlocks blocks;
  float inv_left_m, inv_right_m;   // 1.0f / left.m, 1.0f / right.m (pf_score_given_kernel: mod_u16)
  float* publish;   // device-resident filter: the float32 score the reference publishes (_update_particle_scores), or nullptr
};

constexpr int kWaves = kBlock / 64;

__device__ __forceinline__ void take_smaller(double& d, int& i, double od, int oi) {
  if (od < d || (od == d && oi < i)) {  // lowest index on ties: the first minimum
    d = od;
    i = oi;
  }
}

// (distance^2, index) minimum over the wavefront, every lane ends with the result
__device__ __forceinline__ void wave_argmin(double& d, int& i) {
#pragma unroll
  for (int mask = 32; mask >= 1; mask >>= 1) {
    const double od = __shfl_xor(d, mask, 64);
    const int oi = __shfl_xor(i, mask, 64);
    take_smaller(d, i, od, oi);
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int mask = 32; mask >= 1; mask >>= 1) v += __shfl_xor(v, mask, 64);  // fixed tree: reproducible
  return v;
}

// ---- nearest map point through a uniform grid ------------------------------------------------------------------------
// The reference answers its three nearest-point queries per particle with KD-trees (localiser.py:282-289); scanning the
// whole map per particle, as pf_score_kernel does on its own, is 3 x 34 758 distance evaluations each.  The map's points
// are binned once (host, at create) into square cells of `cell` metres, one index per polyline over a common box:
// `start` [cells + 1] and the points themselves in cell order (x, y, index in the polyline).  A query visits the cell of the particle,
// then ring after ring of cells round it; after ring r every point not yet seen lies outside the block of
// (2r + 1)^2 cells, i.e. at least `margin` = the particle's distance to the nearest side of that block away (sides on
// the box's own border do not count: nothing lies beyond them), so the search stops as soon as the best squared
// distance is below margin^2.  Same float64 distance expression, ties to the lower index: the answer is the
// exhaustive scan's, bit for bit.  A particle whose search is not settled after kMaxRings rings (far from the track,
// or outside the box) is scanned exhaustively by its whole wavefront.
struct GridArgs {
  const float* states;   // [P][3]
  const int* live;       // or nullptr
  Track track[3];
  GridIndex grid[3];
  GridBlocks blocks;
  double x0, y0, cell;
  int nx, ny;
  int32_t* index_out;    // [P][3]
  double* d2_out;        // [P][3]
};
constexpr int kMaxRings = 4;

// (distance^2, index) minimum over the LANES consecutive lanes that share a query (LANES a power of two), every one
// of them ends with the result
template <int LANES>
__device__ __forceinline__ void group_argmin(double& d, int& i) {
#pragma unroll
  for (int mask = LANES / 2; mask >= 1; mask >>= 1) {
    const double od = __shfl_xor(d, mask, 64);
    const int oi = __shfl_xor(i, mask, 64);
    take_smaller(d, i, od, oi);
  }
}

// One run of the cell-ordered arrays - the points of a row of cells - against the query, LANES lanes side by side, two
// points per lane and trip: consecutive lanes read consecutive points, so a trip is a handful of cache lines, where a lane
// walking its own run (rounds 2-4: one lane per query) touched a line per lane and array - the search was bound by the
// number of LINES its gathers asked the vector cache for, not by their bytes (round 5, profiles/r05_pf_sq_counters.json).
template <int LANES>
__device__ __forceinline__ void scan_run(const double* __restrict__ gx, const double* __restrict__ gy,
                                         const int* __restrict__ gi, int j0, int j1, int sub, double px, double py,
                                         double& best, int& best_i) {
  for (int j = j0 + sub; j < j1; j += 2 * LANES) {
    const int k = min(j + LANES, j1 - 1);   // past the run: its last point again (harmless: same distance, same index)
    const double ax = gx[j], ay = gy[j], bx = gx[k], by = gy[k];
    const int ai = gi[j], bi = gi[k];
    const double adx = px - ax, ady = py - ay, bdx = px - bx, bdy = py - by;
    take_smaller(best, best_i, adx * adx + ady * ady, ai);
    take_smaller(best, best_i, bdx * bdx + bdy * bdy, bi);
  }
}

// The nearest point of one polyline to (px, py) through the grid, by the LANES lanes of a group together (`sub` = the
// lane's place in its group; every argument but `sub` is the same for all of them).  The block of 3 x 3 cells round the
// query first - its three rows requested together - then 5 x 5 ... (2 kMaxRings + 1)^2, each block scanned whole; after a
// block every point not yet seen is at least `margin` = the distance to the nearest side of the block that has cells
// beyond it away, so the search stops as soon as the best squared distance is below margin^2.  Returns whether it settled;
// (best, best_i) are the group's result then.  Same float64 distance expression as the exhaustive scan, ties to the lower
// index: its answer, bit for bit.  (The single cell of rounds 2-4 is no longer tried first: the left and right limits
// lie 4.75 m from the centre line, more than half a cell - two queries in three never settled there.)
template <int LANES>
__device__ __forceinline__ bool grid_nearest(double px, double py, const GridGeometry& g, const GridBlocks& blocks,
                                             const GridIndex (&grids)[3], int t, int sub, double& best, int& best_i) {
  // (any cell will do for a correct answer - the margins below are measured from the particle itself - so the particle's
  // cell comes from a product with 1 / cell here, where the host bins the map's points with the division)
  const double inv_cell = 1.0 / g.cell;   // wave-uniform: scalar registers
  const int ix = min(max(static_cast<int>(floor((px - g.x0) * inv_cell)), 0), g.nx - 1);
  const int iy = min(max(static_cast<int>(floor((py - g.y0) * inv_cell)), 0), g.ny - 1);
  for (int r = 1; r <= kMaxRings; ++r) {
    best = INFINITY;
    best_i = 0x7fffffff;
    if (r == 1 && blocks.start != nullptr) {   // the 3 x 3 block: one run
      const int c = t * blocks.cells + iy * g.nx + ix;
      scan_run<LANES>(blocks.x, blocks.y, blocks.index, blocks.start[c], blocks.start[c + 1], sub, px, py, best, best_i);
    } else {
      // (rare from the second ring on: the polyline's own arrays are chosen here, not in front of the search)
      const GridIndex& gr = (t == 0) ? grids[0] : (t == 1) ? grids[1] : grids[2];
      const int x_lo = max(ix - r, 0), x_hi = min(ix + r, g.nx - 1);
      const int y_lo = max(iy - r, 0), y_hi = min(iy + r, g.ny - 1);
      for (int cy = y_lo; cy <= y_hi; ++cy)
        scan_run<LANES>(gr.x, gr.y, gr.index, gr.start[cy * g.nx + x_lo], gr.start[cy * g.nx + x_hi + 1], sub, px, py, best,
                        best_i);
    }
    group_argmin<LANES>(best, best_i);
    // distance to the nearest side of the block that has cells beyond it
    double margin = INFINITY;
    if (ix - r > 0) margin = fmin(margin, px - (g.x0 + (ix - r) * g.cell));
    if (ix + r < g.nx - 1) margin = fmin(margin, (g.x0 + (ix + r + 1) * g.cell) - px);
    if (iy - r > 0) margin = fmin(margin, py - (g.y0 + (iy - r) * g.cell));
    if (iy + r < g.ny - 1) margin = fmin(margin, (g.y0 + (iy + r + 1) * g.cell) - py);
    // (1 - 1e-9: the comparison must hold for the exact values; strictly below, so that a point outside the block at
    // exactly the same distance, which could carry a lower index, is still looked at)
    if (margin > 0.0 && best < margin * margin * (1.0 - 1e-9)) return true;
    if (margin == INFINITY) return true;   // the block covers the whole box
  }
  return false;
}

// The exhaustive scan of one polyline by a whole wavefront (what the grid search falls back to: a query far from the
// track, outside the box, non-finite); every lane ends with the result.
__device__ __forceinline__ void wave_scan(const Track& track, double qx, double qy, int lane, double& d, int& i) {
  d = INFINITY;
  i = 0x7fffffff;
  for (int m = lane; m < track.m; m += 64) {
    const double dx = qx - track.xy[2 * m], dy = qy - track.xy[2 * m + 1];
    const double dd = dx * dx + dy * dy;
    if (dd < d) {   // ascending m per lane: the first minimum stays
      d = dd;
      i = m;
    }
  }
  wave_argmin(d, i);
}

// kGroupLanes lanes per (particle, polyline) query, 64 / kGroupLanes queries per wavefront at a time, kQueriesPerWave per
// wavefront.  (A/B builds: ACMPC_HIPCC_EXTRA=-DACMPC_PF_GROUP_LANES=8)
#ifndef ACMPC_PF_GROUP_LANES
#define ACMPC_PF_GROUP_LANES 16
#endif
constexpr int kGroupLanes = ACMPC_PF_GROUP_LANES;
constexpr int kQueriesPerWave = 16;
static_assert(kGroupLanes == 4 || kGroupLanes == 8 || kGroupLanes == 16 || kGroupLanes == 32, "a power of two, several groups per wave");
static_assert(kQueriesPerWave % (64 / kGroupLanes) == 0, "whole rounds");

__global__ void __launch_bounds__(64) pf_nearest_kernel(const GridArgs a, const int P_arg) {
  const int lane = threadIdx.x;
  const int P = (a.live != nullptr) ? a.live[0] : P_arg;
  const int first = blockIdx.x * kQueriesPerWave;
  if (first >= 3 * P) return;   // wave-uniform
  const int group = lane / kGroupLanes, sub = lane % kGroupLanes;
  const GridGeometry geo{a.x0, a.y0, a.cell, a.nx, a.ny};
#pragma unroll 1
  for (int round = 0; round < kQueriesPerWave / (64 / kGroupLanes); ++round) {
    const int base = first + round * (64 / kGroupLanes);
    if (base >= 3 * P) break;   // wave-uniform
    const int query = base + group;
    const bool active = query < 3 * P;
    const int p = active ? query / 3 : P - 1;
    const int t = active ? query - 3 * p : 0;
    const double px = a.states[3 * p], py = a.states[3 * p + 1];
    double best;
    int best_i;
    const bool settled = grid_nearest<kGroupLanes>(px, py, geo, a.blocks, a.grid, t, sub, best, best_i);
    // the rest: all 64 lanes scan the polyline side by side for each unsettled query of the wave
    unsigned long long pending = __ballot(!settled && sub == 0);
    while (pending != 0ull) {
      const int src = __builtin_amdgcn_readfirstlane(__ffsll(static_cast<long long>(pending)) - 1);
      pending &= pending - 1ull;
      const double qx = __shfl(px, src, 64), qy = __shfl(py, src, 64);
      const int qt = __builtin_amdgcn_readlane(t, src);
      double d;
      int i;
      wave_scan(a.track[qt], qx, qy, lane, d, i);
      if (lane == src) {
        best = d;
        best_i = i;
      }
    }
    if (active && sub == 0) {
      a.index_out[3 * p + t] = best_i;
      a.d2_out[3 * p + t] = best;
    }
  }
}

// One workgroup scores PB particles.  Every map point a thread loads is compared against all PB of them, so the
// map (557 kB for the three 11.6 k-point polylines) crosses the L2 once per workgroup instead of once per particle -
// at 100 000 particles that read was the bound (55 GB per scoring call).  Reductions are wave shuffles plus one
// kWaves-entry exchange through LDS per polyline, for all PB particles at once (the first form ran three 256-wide
// LDS trees of eight barriers each per particle).  PB = 1 keeps one particle per workgroup for the reference's
// particle counts (500: configs/monza.yaml:47), where the launch has to fill the chip with workgroups.
template <int PB>
__global__ void __launch_bounds__(kBlock) pf_score_kernel(const ScoreArgs a, const int P_arg) {
  __shared__ double s_d[3][PB][kWaves];
  __shared__ int s_i[3][PB][kWaves];
  __shared__ double s_sum[PB][kWaves];
  __shared__ float s_frame[PB][4];   // cos, sin of the placement angle and the particle's position, per particle
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int p0 = blockIdx.x * PB;
  if (a.live != nullptr) {   // wave-uniform
    if (p0 >= a.live[0]) return;
  }
  const int P = (a.live != nullptr) ? a.live[0] : P_arg;
  double px[PB], py[PB];
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const int p = min(p0 + q, P - 1);  // the last workgroup repeats its last particle; only p < P is written
    px[q] = a.states[3 * p];
    py[q] = a.states[3 * p + 1];
  }

  // three nearest-neighbour queries (localiser.py:282-289): first minimum of the float64 squared distance - found
  // already by the grid search, or scanned here
  const Track tracks[3] = {a.centre, a.left, a.right};
  const bool own_search = PB == 1 && a.own_grid_search != 0;   // wave-uniform
  if (own_search) {
    // the reference's particle counts (hundreds): a wavefront per polyline, its 64 lanes across the points of the cells
    // round the particle - ~150 points per polyline instead of all 11 600 (rounds 1-4 scanned the whole map here: 35 us
    // for 500 particles, every workgroup reading 556 kB through the L2)
    if (wave < 3) {
      double best;
      int best_i;
      const int which = __builtin_amdgcn_readfirstlane(wave);
      if (!grid_nearest<64>(px[0], py[0], a.geometry, a.blocks, a.grid, which, lane, best, best_i))
        wave_scan(tracks[wave], px[0], py[0], lane, best, best_i);
      if (lane < kWaves) {
        s_d[wave][0][lane] = (lane == 0) ? best : INFINITY;
        s_i[wave][0][lane] = (lane == 0) ? best_i : 0x7fffffff;
      }
    }
  }
  // the placement of the observation in each particle's frame needs cos / sin of ITS angle: once per particle (round 5;
  // every thread used to evaluate both for every particle of the workgroup - a fifth of the kernel's instructions at PB = 8)
  if (tid < PB) {
    const int p = min(p0 + tid, P - 1);
    const float angle = -a.states[3 * p + 2] + 1.57079632679489661923f;
    s_frame[tid][0] = cosf(angle);
    s_frame[tid][1] = sinf(angle);
    s_frame[tid][2] = a.states[3 * p];
    s_frame[tid][3] = a.states[3 * p + 1];
  }
  const bool given = a.given_index != nullptr || own_search;   // wave-uniform
  if (given && !own_search) {
    for (int e = tid; e < 3 * PB * kWaves; e += kBlock) {
      const int w = e % kWaves, q = (e / kWaves) % PB, t = e / (kWaves * PB);
      const int p = min(p0 + q, P - 1);
      s_d[t][q][w] = (w == 0) ? a.given_d2[3 * p + t] : INFINITY;
      s_i[t][q][w] = (w == 0) ? a.given_index[3 * p + t] : 0x7fffffff;
    }
  }
#pragma unroll
  for (int t = 0; t < 3 && !given; ++t) {
    double best[PB];
    int best_i[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      best[q] = INFINITY;
      best_i[q] = 0x7fffffff;
    }
    const double* __restrict__ xy = tracks[t].xy;
    for (int m = tid; m < tracks[t].m; m += kBlock) {
      const double mx = xy[2 * m], my = xy[2 * m + 1];
#pragma unroll
      for (int q = 0; q < PB; ++q) {
        const double dx = px[q] - mx, dy = py[q] - my;
        const double d = dx * dx + dy * dy;
        if (d < best[q]) {  // ascending m per thread: the first minimum stays
          best[q] = d;
          best_i[q] = m;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      wave_argmin(best[q], best_i[q]);
      if (lane == 0) {
        s_d[t][q][wave] = best[q];
        s_i[t][q][wave] = best_i[q];
      }
    }
  }
  __syncthreads();
  int nearest[3][PB];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      double d = s_d[t][q][0];
      int i = s_i[t][q][0];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) take_smaller(d, i, s_d[t][q][w], s_i[t][q][w]);
      nearest[t][q] = (i == 0x7fffffff) ? 0 : i;   // a non-finite position is nearest to nothing: point 0, like np.argmin
    }

  // observation placed in each particle's frame vs the map limits ahead of the nearest points (:330-410)
  // The reference places the observation in float32 (float32 states and observation, :330-353) and only then
  // subtracts the float64 map: do the same, so that the placed points round the way its do.
  const int K = a.k_left + a.k_right;
#pragma unroll
  for (int q = 0; q < PB; ++q) {
    const float ca = s_frame[q][0], sa = s_frame[q][1], px32 = s_frame[q][2], py32 = s_frame[q][3];
    double sum = 0.0;
    for (int k = tid; k < K; k += kBlock) {
      const float ox = a.obs[2 * k], oy = a.obs[2 * k + 1];
      const double wx = (ca * ox + sa * oy) + px32;   // transpose of [[cos, -sin], [sin, cos]] (:355-364)
      const double wy = (-sa * ox + ca * oy) + py32;
      const bool is_left = k < a.k_left;
      const Track t = is_left ? a.left : a.right;
      const int i = is_left ? k : k - a.k_left;
      const int count = is_left ? a.k_left : a.k_right;
      const int closest = is_left ? nearest[1][q] : nearest[2][q];
      // np.linspace(closest, closest + count, count, dtype=uint16): closest + i, except the last entry = closest + count
      const int off = (count > 1 && i == count - 1) ? count : i;
      const int idx = ((closest + off) & 0xffff) % t.m;
      const double dx = wx - t.xy[2 * idx], dy = wy - t.xy[2 * idx + 1];
      sum += sqrt(dx * dx + dy * dy);
    }
    sum = wave_sum(sum);
    if (lane == 0) s_sum[q][wave] = sum;
  }
  __syncthreads();

  if (tid < PB && p0 + tid < P) {
    const int q = tid, p = p0 + tid;
    double total = s_sum[q][0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) total += s_sum[q][w];
    const double error = total / static_cast<double>(K);
    const double phi = a.states[3 * p + 2];
    // nearest[...][q] with a run-time q: read them back from LDS (the registers are indexed statically)
    int i_centre = s_i[0][q][0], i_left = s_i[1][q][0], i_right = s_i[2][q][0];
    double dc = s_d[0][q][0], dl = s_d[1][q][0], dr = s_d[2][q][0];
    for (int w = 1; w < kWaves; ++w) {
      take_smaller(dc, i_centre, s_d[0][q][w], s_i[0][q][w]);
      take_smaller(dl, i_left, s_d[1][q][w], s_i[1][q][w]);
      take_smaller(dr, i_right, s_d[2][q][w], s_i[2][q][w]);
    }
    i_centre = (i_centre == 0x7fffffff) ? 0 : i_centre;
    i_left = (i_left == 0x7fffffff) ? 0 : i_left;
    i_right = (i_right == 0x7fffffff) ? 0 : i_right;
    // heading of the centreline at the nearest point, indices mod (len - 1) (:291-318)
    const int m1 = a.centre.m - 1;
    const int here = i_centre % m1, next = (i_centre + 1) % m1;
    const double track_heading = atan2(a.centre.xy[2 * next + 1] - a.centre.xy[2 * here + 1],
                                       a.centre.xy[2 * next] - a.centre.xy[2 * here]);
    const double raw = track_heading - phi + kPi;
    const double heading = fabs(raw - floor(raw / (2 * kPi)) * (2 * kPi) - kPi);
    const double offset = sqrt(dc);
    const double z = (error - a.mean) / a.sigma;
    const double score = exp(-z * z / 2.0) / sqrt(2.0 * kPi) / a.sigma / a.scale;
    a.track_indices[3 * p] = i_centre;
    a.track_indices[3 * p + 1] = i_left;
    a.track_indices[3 * p + 2] = i_right;
    a.minimum_offset[p] = offset;
    a.heading_offset[p] = heading;
    a.error[p] = error;
    a.score[p] = score;
    if (a.publish != nullptr) a.publish[p] = static_cast<float>(score);
    a.valid[p] = (heading < a.thr_rotation && offset < a.thr_offset && error < a.thr_error) ? 1 : 0;
  }
}

// x mod m for 0 <= x < 65 536 and m >= 1 (the reference's uint16 index arithmetic, :388-392) without the integer division
// the compiler emits for `%`: the quotient from a float32 product (inv_m = 1.0f / m), off by one at most, then put right.
__device__ __forceinline__ int mod_u16(int x, int m, float inv_m) {
  const int q = static_cast<int>(static_cast<float>(x) * inv_m);
  int r = x - q * m;
  r = (r < 0) ? r + m : r;
  return (r >= m) ? r - m : r;
}

// v of lane (l ^ MASK): the vector pipe's own lane permutations where it has one for the mask (inside a quad, across the
// halves of a row of 16), the LDS crossbar otherwise
template <int MASK>
__device__ __forceinline__ double from_lane_xor(double v) {
  if constexpr (MASK == 1 || MASK == 2 || MASK == 8) {
    constexpr int ctrl = (MASK == 1) ? 0xB1 : (MASK == 2) ? 0x4E : 0x128;   // quad_perm [1,0,3,2] / [2,3,0,1] / row_ror:8
    const long long bits = __double_as_longlong(v);
    // (every lane has a source under these controls: `old` is never kept, and passing the value itself saves its zeroing)
    const int lo = __builtin_amdgcn_update_dpp(static_cast<int>(bits), static_cast<int>(bits), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(static_cast<int>(bits >> 32), static_cast<int>(bits >> 32), ctrl, 0xf, 0xf, false);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
  } else {
    return __shfl_xor(v, MASK, 64);
  }
}

// ((wave_sum(s[0]) + wave_sum(s[1])) + wave_sum(s[2])) + wave_sum(s[3]), bit for bit, in ten exchanges instead of twenty-
// four.  wave_sum's tree adds lane l ^ 32, then l ^ 16, ... l ^ 1; after the first stage both halves of the wave hold the
// same 32 partial sums (a + b = b + a), so two sums share a vector from there - and after the second stage four.  Every
// sum is still reduced by exactly wave_sum's tree.
__device__ __forceinline__ double wave_sum_of_four(const double (&s)[4], int lane) {
  double t[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) t[w] = s[w] + from_lane_xor<32>(s[w]);
  double a = (lane < 32) ? t[0] : t[1], b = (lane < 32) ? t[2] : t[3];
  a += from_lane_xor<16>(a);
  b += from_lane_xor<16>(b);
  double c = (lane & 16) ? b : a;   // lanes 0-15: s[0]'s partial sums, 16-31: s[2]'s, 32-47: s[1]'s, 48-63: s[3]'s
  c += from_lane_xor<8>(c);
  c += from_lane_xor<4>(c);
  c += from_lane_xor<2>(c);
  c += from_lane_xor<1>(c);
  auto of_lane = [&](int l) {
    const long long bits = __double_as_longlong(c);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(bits), l), hi = __builtin_amdgcn_readlane(static_cast<int>(bits >> 32), l);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
  };
  return ((of_lane(0) + of_lane(32)) + of_lane(16)) + of_lane(48);
}

// The scoring of particles whose nearest points are GIVEN (pf_nearest_kernel ran in front: 4 096 particles and more), one
// WAVEFRONT for PW particles and no workgroup at all: lane l (< PW) holds particle l's frame and nearest points and
// evaluates its score at the end; in between the 64 lanes place the observation in one particle's frame after the other
// (wave-uniform frame: read out of lane l's registers) and sum the distances.  pf_score_kernel<8> spent its time waiting -
// two barriers and four dependent trips to memory per workgroup of 3 us of work, 1 150 instructions per wave (profiles/
// r05_pf_sq_counters.json) - and evaluated the eight tails (atan2, exp, float64 divisions) on eight lanes of one wave.
// Same bits: pf_score_kernel's thread `tid` sums the observation points k = tid, tid + 256, ...; its wave w reduces them
// with wave_sum and the four sums are added in the order of the waves.  Here lane l does the same for each of the four
// slots w (k = 64 w + l, + 256, ...), each slot is reduced by the same tree, and they are added in the same order.
// ONE_TRIP: at most 256 observation points (the reference's downsampled observation: 200), a point per lane and slot.
// NO_WRAP: nearest index + offset stays below 65 536 and below twice the polyline's length on both sides (launch_score
// checks the lengths), so the uint16 wrap never happens and the modulo is one conditional subtraction.
// PW (1 ... 64, wave-uniform): particles per wave - launch_score picks it so that the launch is whole generations of waves.
template <bool ONE_TRIP, bool NO_WRAP>
__global__ void __launch_bounds__(64) pf_score_given_kernel(const ScoreArgs a, const int P_arg, const int PW) {
  static_assert(kWaves == 4, "wave_sum_of_four");
  const int lane = threadIdx.x;
  const int p0 = blockIdx.x * PW;
  const int P = (a.live != nullptr) ? a.live[0] : P_arg;
  if (p0 >= P) return;   // wave-uniform
  // phase 0: lane l's own particle (lanes beyond the count repeat the last one and write nothing)
  const int p_own = min(p0 + min(lane, PW - 1), P - 1);
  const float sx = a.states[3 * p_own], sy = a.states[3 * p_own + 1], sphi = a.states[3 * p_own + 2];
  const float angle = -sphi + 1.57079632679489661923f;
  const float ca_own = cosf(angle), sa_own = sinf(angle);
  int i_centre = a.given_index[3 * p_own], i_left = a.given_index[3 * p_own + 1], i_right = a.given_index[3 * p_own + 2];
  const double dc = a.given_d2[3 * p_own];
  i_centre = (i_centre == 0x7fffffff) ? 0 : i_centre;   // a non-finite position is nearest to nothing: point 0, like np.argmin
  i_left = (i_left == 0x7fffffff) ? 0 : i_left;
  i_right = (i_right == 0x7fffffff) ? 0 : i_right;

  // phase 1: the observation against the map limits ahead of each particle's nearest points (:330-410), float32 placement
  // as the reference's, float64 from the subtraction of the map on
  const int K = a.k_left + a.k_right;
  // what observation point k contributes for every particle alike
  struct Slot {
    float ox, oy;
    int off;        // np.linspace(closest, closest + count, count, dtype=uint16): closest + i, except the last entry = closest + count
    bool is_left, live;
  };
  auto slot_of = [&](int k_raw) {
    Slot t;
    t.live = k_raw < K;
    const int k = min(k_raw, K - 1);   // (a lane past the end evaluates the last point and adds nothing)
    t.ox = a.obs[2 * k];
    t.oy = a.obs[2 * k + 1];
    t.is_left = k < a.k_left;
    const int i = t.is_left ? k : k - a.k_left;
    const int count = t.is_left ? a.k_left : a.k_right;
    t.off = (count > 1 && i == count - 1) ? count : i;
    return t;
  };
  // in two halves, so that a particle's four map points are requested together and not one after the other's arrival
  auto map_point = [&](const Slot& t, int near_left, int near_right) {
    const double* __restrict__ xy = t.is_left ? a.left.xy : a.right.xy;
    const int m = t.is_left ? a.left.m : a.right.m;
    const int ahead = (t.is_left ? near_left : near_right) + t.off;
    int idx;
    if constexpr (NO_WRAP) {
      idx = (ahead >= m) ? ahead - m : ahead;
    } else {
      idx = mod_u16(ahead & 0xffff, m, t.is_left ? a.inv_left_m : a.inv_right_m);
    }
    return *reinterpret_cast<const double2*>(xy + 2 * idx);   // (x, y): 16-byte aligned
  };
  auto distance = [&](const Slot& t, float ca, float sa, float px32, float py32, double2 at) {
    const double wx = (ca * t.ox + sa * t.oy) + px32;   // transpose of [[cos, -sin], [sin, cos]] (:355-364)
    const double wy = (-sa * t.ox + ca * t.oy) + py32;
    const double dx = wx - at.x, dy = wy - at.y;
    return sqrt(dx * dx + dy * dy);
  };
  Slot slots[kWaves];
#pragma unroll
  for (int w = 0; w < kWaves; ++w) slots[w] = slot_of(64 * w + lane);
  double total_own = 0.0;
  const int count_here = min(PW, P - p0);
#pragma unroll 1
  for (int q = 0; q < count_here; ++q) {
    const float ca = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ca_own), q));
    const float sa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sa_own), q));
    const float px32 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), q));
    const float py32 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), q));
    const int near_left = __builtin_amdgcn_readlane(i_left, q), near_right = __builtin_amdgcn_readlane(i_right, q);
    double sum[kWaves];
    double2 at[kWaves];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) at[w] = map_point(slots[w], near_left, near_right);
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise reuses one register quad and serialises the four loads)
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const double d = distance(slots[w], ca, sa, px32, py32, at[w]);
      sum[w] = slots[w].live ? d : 0.0;
    }
    if constexpr (!ONE_TRIP) {
      for (int base = kBlock; base < K; base += kBlock) {
        Slot t[kWaves];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          t[w] = slot_of(base + 64 * w + lane);
          at[w] = map_point(t[w], near_left, near_right);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
          const double d = distance(t[w], ca, sa, px32, py32, at[w]);
          sum[w] += t[w].live ? d : 0.0;
        }
      }
    }
    const double total = wave_sum_of_four(sum, lane);
    total_own = (lane == q) ? total : total_own;
  }

  // phase 2: lane l finishes particle l (heading of the centre line at the nearest point, score, validity)
  if (lane < count_here) {
    const int p = p0 + lane;
    const double error = total_own / static_cast<double>(K);
    const double phi = sphi;
    // heading of the centreline at the nearest point, indices mod (len - 1) (:291-318)
    const int m1 = a.centre.m - 1;
    const int here = i_centre % m1, next = (i_centre + 1) % m1;
    const double track_heading = atan2(a.centre.xy[2 * next + 1] - a.centre.xy[2 * here + 1],
                                       a.centre.xy[2 * next] - a.centre.xy[2 * here]);
    const double raw = track_heading - phi + kPi;
    const double heading = fabs(raw - floor(raw / (2 * kPi)) * (2 * kPi) - kPi);
    const double offset = sqrt(dc);
    const double z = (error - a.mean) / a.sigma;
    const double score = exp(-z * z / 2.0) / sqrt(2.0 * kPi) / a.sigma / a.scale;
    a.track_indices[3 * p] = i_centre;
    a.track_indices[3 * p + 1] = i_left;
    a.track_indices[3 * p + 2] = i_right;
    a.minimum_offset[p] = offset;
    a.heading_offset[p] = heading;
    a.error[p] = error;
    a.score[p] = score;
    if (a.publish != nullptr) a.publish[p] = static_cast<float>(score);
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
                                                             double* out /*[5]: x, y, yaw, max_dist, max_angle*/,
                                                             const int* live = nullptr, int* counts_out = nullptr) {
  __shared__ double s[4][kBlock];
  const int tid = threadIdx.x;
  if (live != nullptr) P = live[0];
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
    if (counts_out != nullptr && live != nullptr)   // (the device-resident filter: its four counts beside the estimate)
      for (int e = 0; e < 4; ++e) counts_out[e] = live[e];
    out[0] = est[0];
    out[1] = est[1];
    out[2] = est[2];
    out[3] = s[0][0];
    out[4] = s[1][0];
  }
}


// ---- device-resident filter (one host round trip per update) ----------------------------------------------------------
// The reference's update cycle (localiser.py:41-77,234-239,420-579) with its random part restated on counter-based
// draws, so that it can run where the particles live:
//   step      every particle moves with its own noisy control (yaw noise on the tyre angle, |velocity + noise|)
//   update    score -> keep the valid particles in order -> too few left: reset along the centre line; otherwise top up
//             to the desired count with copies of kept particles drawn in proportion to their score (inverse CDF on a
//             prefix sum) plus Gaussian noise -> score-weighted estimate and the two convergence numbers
// The reference draws from NumPy's global Mersenne Twister in a fixed call order, which a parallel kernel cannot
// follow; `particle_filter.ParticleFilter` (NumPy resampling, the reference's draw order) therefore stays the parity
// mode, and this path is pinned by its own restatement (oracle pf_resample_counter_based): the weights are integers
// (floor(score * 2^40)), so the prefix sums, the draws (mulhi of a 64-bit Philox word with the total) and the picked
// indices are exact in any summation order.
constexpr int kScanBlock = 1024;
constexpr uint32_t kTagResample = 0x52534d50u;  // "RSMP"
constexpr uint32_t kTagControl = 0x4354524cu;   // "CTRL"

struct FilterArgs {
  const float* states_in;   // [n_live][3]
  const float* scores_in;   // [n_live]  (float32 scores of this update)
  const double* score;      // [n_live]  float64 scores of this update
  const uint8_t* valid;     // [n_live]
  float* states_out;        // [capacity][3]
  float* scores_out;        // [capacity]
  unsigned long long* cdf;  // [n_live] workspace: inclusive prefix sums of the kept weights
  int* kept;                // [n_live] workspace: source index of kept particle k
  int* counts;              // [4]: n_live (in/out), n_valid (out), was_reset (out), spare
  const double* centre;     // map centre line [m][2] (reset pattern)
  int m_centre;
  int capacity;             // max_particles
  int n_desired, minimum_particles;
  double sigma_x, sigma_y, sigma_yaw;
  uint32_t seed_lo, seed_hi, counter;
};

__device__ __forceinline__ unsigned long long weight_of(double score) {
  // floor(score * 2^40) for finite positive scores, 0 otherwise (NaN, negative); scores are <= 1 by construction
  if (!(score > 0.0)) return 0ull;
  const double scaled = score * 1099511627776.0;
  return scaled >= 1.8446744073709552e19 ? 0xffffffffffffffffull : static_cast<unsigned long long>(scaled);
}

// exclusive prefix sum of one value per thread over the workgroup (kScanBlock threads); returns the offset, *total
__device__ unsigned long long block_exclusive_scan(unsigned long long v, unsigned long long* s_scan,
                                                   unsigned long long* total) {
  const int tid = threadIdx.x;
  s_scan[tid] = v;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    const unsigned long long add = (tid >= off) ? s_scan[tid - off] : 0ull;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  *total = s_scan[kScanBlock - 1];
  const unsigned long long inclusive = s_scan[tid];
  __syncthreads();
  return inclusive - v;
}

__global__ void __launch_bounds__(kScanBlock) pf_resample_kernel(const FilterArgs a) {
  __shared__ unsigned long long s_scan[kScanBlock];
  const int tid = threadIdx.x;
  const int n = a.counts[0];
  const int chunk = (n + kScanBlock - 1) / kScanBlock;
  const int begin = min(tid * chunk, n), end = min(begin + chunk, n);
  // 1. stable compaction of the valid particles
  unsigned long long local = 0ull, total = 0ull;
  for (int p = begin; p < end; ++p) local += a.valid[p] ? 1ull : 0ull;
  unsigned long long offset = block_exclusive_scan(local, s_scan, &total);
  const int n_valid = static_cast<int>(total);
  for (int p = begin; p < end; ++p)
    if (a.valid[p]) a.kept[offset++] = p;
  __syncthreads();
  __threadfence_block();
  if (n_valid < a.minimum_particles) {
    // _reset_filter (localiser.py:468-485): spread evenly along the centre line, heading along it, uniform scores
    const int count = a.capacity;
    for (int k = tid; k < count; k += kScanBlock) {
      // np.linspace(0, m - 3, count).astype(int32)
      const double pos = (count > 1) ? static_cast<double>(k) * (static_cast<double>(a.m_centre - 3) / static_cast<double>(count - 1))
                                     : 0.0;
      int idx = static_cast<int>((k == count - 1 && count > 1) ? static_cast<double>(a.m_centre - 3) : pos);
      const double x = a.centre[2 * idx], y = a.centre[2 * idx + 1];
      const double yaw = atan2(a.centre[2 * (idx + 1) + 1] - y, a.centre[2 * (idx + 1)] - x);
      a.states_out[3 * k] = static_cast<float>(x);
      a.states_out[3 * k + 1] = static_cast<float>(y);
      a.states_out[3 * k + 2] = static_cast<float>(yaw);
      a.scores_out[k] = 1.0f / static_cast<float>(count);
    }
    if (tid == 0) {
      a.counts[0] = count;
      a.counts[1] = n_valid;
      a.counts[2] = 1;
    }
    return;
  }
  // 2. integer weights of the kept particles and their inclusive prefix sums (exact: order does not matter)
  const int kchunk = (n_valid + kScanBlock - 1) / kScanBlock;
  const int kb = min(tid * kchunk, n_valid), ke = min(kb + kchunk, n_valid);
  local = 0ull;
  for (int k = kb; k < ke; ++k) local += weight_of(a.score[a.kept[k]]);
  unsigned long long wtotal = 0ull;
  offset = block_exclusive_scan(local, s_scan, &wtotal);
  const bool uniform = wtotal == 0ull;   // all scores zero / NaN: uniform weights (localiser.py:523-526)
  if (uniform) {
    offset = static_cast<unsigned long long>(kb);
    wtotal = static_cast<unsigned long long>(n_valid);
  }
  for (int k = kb; k < ke; ++k) {
    offset += uniform ? 1ull : weight_of(a.score[a.kept[k]]);
    a.cdf[k] = offset;
  }
  __syncthreads();
  __threadfence_block();
  // 3. kept particles first, in order
  for (int k = tid; k < n_valid; k += kScanBlock) {
    const int p = a.kept[k];
    a.states_out[3 * k] = a.states_in[3 * p];
    a.states_out[3 * k + 1] = a.states_in[3 * p + 1];
    a.states_out[3 * k + 2] = a.states_in[3 * p + 2];
    a.scores_out[k] = a.scores_in[p];
  }
  // 4. top up: new particle j = kept[upper_bound(cdf, mulhi(r, total))] + noise
  const int n_new = max(0, min(a.n_desired, a.capacity) - n_valid);
  const uint32_t key[2] = {a.seed_lo, a.seed_hi};
  for (int j = tid; j < n_new; j += kScanBlock) {
    const uint32_t c_pick[4] = {static_cast<uint32_t>(j), a.counter, kTagResample, 0u};
    const uint32_t c_noise[4] = {static_cast<uint32_t>(j), a.counter, kTagResample, 1u};
    uint32_t r[4], q[4];
    acmpc::philox4x32_10(c_pick, key, r);
    acmpc::philox4x32_10(c_noise, key, q);
    const unsigned long long word = (static_cast<unsigned long long>(r[0]) << 32) | r[1];
    const unsigned long long target = __umul64hi(word, wtotal);   // uniform in [0, total)
    int lo = 0, hi = n_valid - 1;                                   // first k with cdf[k] > target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a.cdf[mid] > target) hi = mid; else lo = mid + 1;
    }
    const int p = a.kept[lo];
    float z0, z1, z2, z3;
    acmpc::box_muller(acmpc::uniform_open(q[0]), acmpc::uniform_open(q[1]), z0, z1);
    acmpc::box_muller(acmpc::uniform_open(q[2]), acmpc::uniform_open(q[3]), z2, z3);
    (void)z3;
    const int k = n_valid + j;
    a.states_out[3 * k] = static_cast<float>(static_cast<double>(a.states_in[3 * p]) + a.sigma_x * static_cast<double>(z0));
    a.states_out[3 * k + 1] =
        static_cast<float>(static_cast<double>(a.states_in[3 * p + 1]) + a.sigma_y * static_cast<double>(z1));
    a.states_out[3 * k + 2] =
        static_cast<float>(static_cast<double>(a.states_in[3 * p + 2]) + a.sigma_yaw * static_cast<double>(z2));
    a.scores_out[k] = a.scores_in[p];
  }
  if (tid == 0) {
    a.counts[0] = n_valid + n_new;
    a.counts[1] = n_valid;
    a.counts[2] = 0;
  }
}

// ---- the same update for many particles (capacity >= kWideFilter): pf_resample_kernel and pf_estimate_kernel are ONE
// workgroup each, which is what the reference's hundreds of particles want (8 us) and what 100 000 do not (607 + 214 us: a
// thread walks ~100 particles of its own, a cache line apart from its neighbour's).  Here every step is a launch over tiles
// of kScanBlock particles, a particle per thread:
//   tiles    per tile: how many valid particles, the sum of their integer weights
//   plan     one workgroup: exclusive prefix sums of the tiles' counts and weights; reset or not; the new counts
//   scatter  per tile: the kept particles' places (stable: tile offset + rank inside the tile) and their inclusive weight sums
//   emit     a thread per output particle: kept ones copied in order, new ones drawn (same draws, same binary search)
// The integers - ranks, prefix sums, picks - are those of the one-workgroup kernel exactly; so are the particles.
constexpr int kWideFilter = 8192;

struct WideScratch {
  unsigned long long* tile_count;    // [tiles]      valid particles per tile, then (plan) their exclusive prefix sums
  unsigned long long* tile_weight;   // [tiles]      likewise the integer weights
  unsigned long long* meta;          // [6]: n before the update, n_valid, total weight, uniform (0 / 1), reset (0 / 1), n_new
};

__global__ void __launch_bounds__(kScanBlock) pf_resample_tiles_kernel(const FilterArgs a, const WideScratch w) {
  __shared__ unsigned long long s_scan[kScanBlock];
  const int n = a.counts[0];
  const int p = blockIdx.x * kScanBlock + threadIdx.x;
  const bool keep = p < n && a.valid[p] != 0;
  unsigned long long count = 0ull, weight = 0ull;
  (void)block_exclusive_scan(keep ? 1ull : 0ull, s_scan, &count);
  (void)block_exclusive_scan(keep ? weight_of(a.score[p]) : 0ull, s_scan, &weight);
  if (threadIdx.x == 0) {
    w.tile_count[blockIdx.x] = count;
    w.tile_weight[blockIdx.x] = weight;
  }
}

// (tiles <= kScanBlock: capacities up to 2^20 particles; beyond, the one-workgroup kernels serve)
__global__ void __launch_bounds__(kScanBlock) pf_resample_plan_kernel(const FilterArgs a, const WideScratch w, const int tiles) {
  __shared__ unsigned long long s_scan[kScanBlock];
  const int tid = threadIdx.x;
  const unsigned long long count = tid < tiles ? w.tile_count[tid] : 0ull;
  const unsigned long long weight = tid < tiles ? w.tile_weight[tid] : 0ull;
  unsigned long long n_valid = 0ull, total = 0ull;
  const unsigned long long count_before = block_exclusive_scan(count, s_scan, &n_valid);
  const unsigned long long weight_before = block_exclusive_scan(weight, s_scan, &total);
  if (tid < tiles) {
    w.tile_count[tid] = count_before;
    w.tile_weight[tid] = weight_before;
  }
  if (tid == 0) {
    const bool reset = static_cast<int>(n_valid) < a.minimum_particles;
    const bool uniform = total == 0ull;   // all scores zero / NaN: uniform weights (localiser.py:523-526)
    const int n_new = reset ? 0 : max(0, min(a.n_desired, a.capacity) - static_cast<int>(n_valid));
    w.meta[0] = static_cast<unsigned long long>(a.counts[0]);
    w.meta[1] = n_valid;
    w.meta[2] = uniform ? n_valid : total;
    w.meta[3] = uniform ? 1ull : 0ull;
    w.meta[4] = reset ? 1ull : 0ull;
    w.meta[5] = static_cast<unsigned long long>(n_new);
    a.counts[0] = reset ? a.capacity : static_cast<int>(n_valid) + n_new;
    a.counts[1] = static_cast<int>(n_valid);
    a.counts[2] = reset ? 1 : 0;
  }
}

__global__ void __launch_bounds__(kScanBlock) pf_resample_scatter_kernel(const FilterArgs a, const WideScratch w) {
  __shared__ unsigned long long s_scan[kScanBlock];
  if (w.meta[4] != 0ull) return;   // reset: nothing is kept
  const int n = static_cast<int>(w.meta[0]);
  const bool uniform = w.meta[3] != 0ull;
  const int p = blockIdx.x * kScanBlock + threadIdx.x;
  const bool keep = p < n && a.valid[p] != 0;
  const unsigned long long weight = keep ? weight_of(a.score[p]) : 0ull;
  unsigned long long unused = 0ull;
  const unsigned long long rank = block_exclusive_scan(keep ? 1ull : 0ull, s_scan, &unused);
  const unsigned long long before = block_exclusive_scan(weight, s_scan, &unused);
  if (keep) {
    const unsigned long long k = w.tile_count[blockIdx.x] + rank;
    a.kept[k] = p;
    a.cdf[k] = uniform ? k + 1ull : w.tile_weight[blockIdx.x] + before + weight;
  }
}

__global__ void __launch_bounds__(256) pf_resample_emit_kernel(const FilterArgs a, const WideScratch w) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (w.meta[4] != 0ull) {
    // _reset_filter (localiser.py:468-485): spread evenly along the centre line, heading along it, uniform scores
    const int count = a.capacity;
    if (idx >= count) return;
    const int k = idx;
    const double pos = (count > 1) ? static_cast<double>(k) * (static_cast<double>(a.m_centre - 3) / static_cast<double>(count - 1))
                                   : 0.0;
    const int at = static_cast<int>((k == count - 1 && count > 1) ? static_cast<double>(a.m_centre - 3) : pos);
    const double x = a.centre[2 * at], y = a.centre[2 * at + 1];
    const double yaw = atan2(a.centre[2 * (at + 1) + 1] - y, a.centre[2 * (at + 1)] - x);
    a.states_out[3 * k] = static_cast<float>(x);
    a.states_out[3 * k + 1] = static_cast<float>(y);
    a.states_out[3 * k + 2] = static_cast<float>(yaw);
    a.scores_out[k] = 1.0f / static_cast<float>(count);
    return;
  }
  const int n_valid = static_cast<int>(w.meta[1]), n_new = static_cast<int>(w.meta[5]);
  const unsigned long long wtotal = w.meta[2];
  if (idx < n_valid) {   // kept particles first, in order
    const int p = a.kept[idx];
    a.states_out[3 * idx] = a.states_in[3 * p];
    a.states_out[3 * idx + 1] = a.states_in[3 * p + 1];
    a.states_out[3 * idx + 2] = a.states_in[3 * p + 2];
    a.scores_out[idx] = a.scores_in[p];
  }
  if (idx < n_new) {   // top up: new particle j = kept[upper_bound(cdf, mulhi(r, total))] + noise
    const int j = idx;
    const uint32_t key[2] = {a.seed_lo, a.seed_hi};
    const uint32_t c_pick[4] = {static_cast<uint32_t>(j), a.counter, kTagResample, 0u};
    const uint32_t c_noise[4] = {static_cast<uint32_t>(j), a.counter, kTagResample, 1u};
    uint32_t r[4], q[4];
    acmpc::philox4x32_10(c_pick, key, r);
    acmpc::philox4x32_10(c_noise, key, q);
    const unsigned long long word = (static_cast<unsigned long long>(r[0]) << 32) | r[1];
    const unsigned long long target = __umul64hi(word, wtotal);   // uniform in [0, total)
    int lo = 0, hi = n_valid - 1;                                   // first k with cdf[k] > target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a.cdf[mid] > target) hi = mid; else lo = mid + 1;
    }
    const int p = a.kept[lo];
    float z0, z1, z2, z3;
    acmpc::box_muller(acmpc::uniform_open(q[0]), acmpc::uniform_open(q[1]), z0, z1);
    acmpc::box_muller(acmpc::uniform_open(q[2]), acmpc::uniform_open(q[3]), z2, z3);
    (void)z3;
    const int k = n_valid + j;
    a.states_out[3 * k] = static_cast<float>(static_cast<double>(a.states_in[3 * p]) + a.sigma_x * static_cast<double>(z0));
    a.states_out[3 * k + 1] =
        static_cast<float>(static_cast<double>(a.states_in[3 * p + 1]) + a.sigma_y * static_cast<double>(z1));
    a.states_out[3 * k + 2] =
        static_cast<float>(static_cast<double>(a.states_in[3 * p + 2]) + a.sigma_yaw * static_cast<double>(z2));
    a.scores_out[k] = a.scores_in[p];
  }
}

// The estimate of pf_estimate_kernel for many particles: partial sums per workgroup of kEstimateTile particles (float64, a
// fixed order: thread, then the tree of the workgroup, then the workgroups in turn), the spread against the estimate in a
// second launch, its maximum in a third.
constexpr int kEstimateTile = 4096;

__global__ void __launch_bounds__(kBlock) pf_estimate_partial_kernel(const float* states, const float* scores, const int* live,
                                                                     double* partial /*[tiles][8]*/) {
  __shared__ double s[7][kBlock];
  const int tid = threadIdx.x;
  const int P = live[0];
  const int begin = blockIdx.x * kEstimateTile, end = min(begin + kEstimateTile, P);
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};   // sum(state * score) x 3, sum(score), sum(state) x 3
  for (int p = begin + tid; p < end; p += kBlock) {
    const double wgt = scores[p];
    for (int c = 0; c < 3; ++c) {
      const double v = static_cast<double>(states[3 * p + c]);
      acc[c] += v * wgt;
      acc[4 + c] += v;
    }
    acc[3] += wgt;
  }
  for (int c = 0; c < 7; ++c) s[c][tid] = acc[c];
  __syncthreads();
  for (int half = kBlock / 2; half > 0; half >>= 1) {
    if (tid < half)
      for (int c = 0; c < 7; ++c) s[c][tid] += s[c][tid + half];
    __syncthreads();
  }
  if (tid < 7) partial[8 * blockIdx.x + tid] = s[tid][0];
}

__device__ __forceinline__ void estimate_from_partials(const double* partial, int tiles, int P, double est[3]) {
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int t = 0; t < tiles; ++t)
    for (int c = 0; c < 7; ++c) acc[c] += partial[8 * t + c];
  for (int c = 0; c < 3; ++c) est[c] = acc[c] / acc[3];
  if (est[0] != est[0] || est[1] != est[1] || est[2] != est[2])   // NaN: uniform weights
    for (int c = 0; c < 3; ++c) est[c] = acc[4 + c] / static_cast<double>(P);
}

__global__ void __launch_bounds__(kBlock) pf_estimate_spread_kernel(const float* states, const int* live, const double* partial,
                                                                    double* spread /*[tiles][2]*/) {
  __shared__ double s[2][kBlock];
  __shared__ double s_est[3];
  const int tid = threadIdx.x;
  const int P = live[0];
  const int tiles = (P + kEstimateTile - 1) / kEstimateTile;
  if (tid == 0) {
    double est[3];
    estimate_from_partials(partial, tiles, P, est);
    for (int c = 0; c < 3; ++c) s_est[c] = est[c];
  }
  __syncthreads();
  const double ex = s_est[0], ey = s_est[1], ez = s_est[2];
  const int begin = blockIdx.x * kEstimateTile, end = min(begin + kEstimateTile, P);
  double md = 0.0, ma = 0.0;
  for (int p = begin + tid; p < end; p += kBlock) {
    const double dx = states[3 * p] - ex, dy = states[3 * p + 1] - ey;
    md = fmax(md, sqrt(dx * dx + dy * dy));
    ma = fmax(ma, fabs(states[3 * p + 2] - ez));
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
    spread[2 * blockIdx.x] = s[0][0];
    spread[2 * blockIdx.x + 1] = s[1][0];
  }
}

__global__ void __launch_bounds__(kBlock) pf_estimate_final_kernel(const int* live, const double* partial, const double* spread,
                                                                   double* out /*[5]*/, int* counts_out) {
  __shared__ double s[2][kBlock];
  const int tid = threadIdx.x;
  const int P = live[0];
  const int tiles = (P + kEstimateTile - 1) / kEstimateTile;
  double md = 0.0, ma = 0.0;
  for (int t = tid; t < tiles; t += kBlock) {
    md = fmax(md, spread[2 * t]);
    ma = fmax(ma, spread[2 * t + 1]);
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
    double est[3];
    estimate_from_partials(partial, tiles, P, est);
    if (counts_out != nullptr)
      for (int e = 0; e < 4; ++e) counts_out[e] = live[e];
    out[0] = est[0];
    out[1] = est[1];
    out[2] = est[2];
    out[3] = s[0][0];
    out[4] = s[1][0];
  }
}

// Localiser.step (localiser.py:41-77): delta = tyre_angle + N(0, sigma_yaw), speed = |velocity + N(0, sigma_v)| per
// particle from Philox (counter = particle, step), then the float32 kinematic step of pf_advance_kernel
__global__ void pf_step_kernel(float* states, const int* counts, float tyre_angle, float velocity, float sigma_yaw,
                               float sigma_v, float wheelbase, float dt, uint32_t seed_lo, uint32_t seed_hi,
                               uint32_t counter) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= counts[0]) return;
  const uint32_t ctr[4] = {static_cast<uint32_t>(p), counter, kTagControl, 0u};
  const uint32_t key[2] = {seed_lo, seed_hi};
  uint32_t r[4];
  acmpc::philox4x32_10(ctr, key, r);
  float z0, z1;
  acmpc::box_muller(acmpc::uniform_open(r[0]), acmpc::uniform_open(r[1]), z0, z1);
  const float delta = tyre_angle + sigma_yaw * z0;
  const float v = fabsf(velocity + sigma_v * z1);
  const float phi = states[3 * p + 2];
  states[3 * p] += (v * cosf(phi)) * dt;
  states[3 * p + 1] += (v * sinf(phi)) * dt;
  states[3 * p + 2] += (v * tanf(delta) / wheelbase) * dt;
}


thread_local std::string g_pf_create_error;

}  // namespace

struct acmpc_pf {
  acmpc_pf_params prm{};
  bool no_grid = false;   // ACMPC_PF_NO_GRID (A/B switch of the tests): read from the environment once, by acmpc_pf_create
  bool workgroup_score = false;   // ACMPC_PF_WORKGROUP_SCORE (likewise): pf_score_kernel<8> behind the grid search, as in rounds 2-5
  bool narrow_filter = false;     // ACMPC_PF_NARROW_FILTER (likewise): the device-resident filter's one-workgroup kernels at any capacity
  int given_wave_slots[3] = {0, 0, 0};   // waves of pf_score_given_kernel's three forms the device holds at once (ensure_device)
  std::vector<double> h_track[3];
  double scale = 1.0;
  bool device_ready = false;
  double* d_track[3] = {nullptr, nullptr, nullptr};
  // the grid over the map's points (see pf_nearest_kernel): built on the host at create, uploaded with the polylines
  std::vector<int> h_grid_start[3], h_grid_index[3];
  std::vector<double> h_grid_x[3], h_grid_y[3];
  double grid_x0 = 0.0, grid_y0 = 0.0, grid_cell = 0.0;
  int grid_nx = 0, grid_ny = 0;
  int* d_grid_start[3] = {nullptr, nullptr, nullptr};
  int* d_grid_index[3] = {nullptr, nullptr, nullptr};
  double* d_grid_x[3] = {nullptr, nullptr, nullptr};
  double* d_grid_y[3] = {nullptr, nullptr, nullptr};
  // the 3 x 3 blocks as single runs (GridBlocks), all three polylines in one set of arrays
  bool no_blocks = false;   // ACMPC_PF_NO_BLOCKS (A/B switch): the first ring row by row from the cell-ordered arrays
  std::vector<int> h_block_start, h_block_index;
  std::vector<double> h_block_x, h_block_y;
  int* d_block_start = nullptr;
  int* d_block_index = nullptr;
  double* d_block_x = nullptr;
  double* d_block_y = nullptr;
  int32_t* d_near_index = nullptr;   // [max_particles][3]
  double* d_near_d2 = nullptr;       // [max_particles][3]
  hipStream_t stream = nullptr;
  // staging for the host-pointer entry points: ONE pinned block up (states | observation) and ONE down (all
  // per-particle results) per scoring call - at the reference's 500 particles the call is a handful of microseconds
  // of kernel and otherwise transfer round trips (it used to make nine of them)
  unsigned char* h_up = nullptr;    // pinned
  unsigned char* h_down = nullptr;  // pinned
  unsigned char* d_down = nullptr;  // [4 P doubles | 3 P int32 | P bytes]
  float* d_states = nullptr;        // [P][3] then the observation [K][2] directly behind it
  float* d_obs = nullptr;
  float* d_aux = nullptr;      // 2 * max_particles floats: delta / velocity, or scores
  double* d_out = nullptr;     // 8 doubles: result of the estimate kernel
  // device-resident filter: ping-pong particle buffers, this update's results, workspace of the resampling
  bool filter_ready = false;
  float* f_states[2] = {nullptr, nullptr};  // [max_particles][3] (+ room for the observation behind buffer 0 / 1)
  float* f_scores[2] = {nullptr, nullptr};
  int f_cur = 0;
  float* f_obs = nullptr;                   // [K][2]
  unsigned long long* f_cdf = nullptr;
  int* f_kept = nullptr;
  int* f_counts = nullptr;                  // [4] n_live, n_valid, was_reset
  unsigned long long* f_wide = nullptr;     // capacity >= kWideFilter: [2 tiles + 6] scratch of the tiled resampling
  double* f_partial = nullptr;              //   ... and of the tiled estimate: [tiles][8] sums, [tiles][2] spreads
  double* h_result = nullptr;               // pinned [8 doubles + 4 ints]
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

// Bins the three polylines into one grid geometry: cells of kGridCell metres over the box of all points (fewer, larger
// cells when that would be more than kGridMaxCells).  Non-finite points are left out (the exhaustive scan never picks
// them either: a NaN distance is never smaller); a map without a finite point gets no grid.
constexpr double kGridCell = 8.0;
constexpr long long kGridMaxCells = 1 << 20;

void build_grid(acmpc_pf* h) {
  double lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
  for (int t = 0; t < 3; ++t)
    for (size_t m = 0; m + 1 < h->h_track[t].size(); m += 2) {
      const double x = h->h_track[t][m], y = h->h_track[t][m + 1];
      if (!std::isfinite(x) || !std::isfinite(y)) continue;
      lo[0] = std::min(lo[0], x);
      hi[0] = std::max(hi[0], x);
      lo[1] = std::min(lo[1], y);
      hi[1] = std::max(hi[1], y);
    }
  h->grid_nx = h->grid_ny = 0;
  if (!(lo[0] <= hi[0])) return;
  double cell = kGridCell;
  for (;;) {
    const double nx = std::floor((hi[0] - lo[0]) / cell) + 1.0, ny = std::floor((hi[1] - lo[1]) / cell) + 1.0;
    if (nx * ny <= static_cast<double>(kGridMaxCells)) {
      h->grid_nx = static_cast<int>(nx);
      h->grid_ny = static_cast<int>(ny);
      break;
    }
    cell *= 2.0;
  }
  h->grid_x0 = lo[0];
  h->grid_y0 = lo[1];
  h->grid_cell = cell;
  const size_t cells = static_cast<size_t>(h->grid_nx) * h->grid_ny;
  for (int t = 0; t < 3; ++t) {
    const size_t M = h->h_track[t].size() / 2;
    std::vector<int> cell_of(M, -1);
    std::vector<int>& start = h->h_grid_start[t];
    start.assign(cells + 1, 0);
    for (size_t m = 0; m < M; ++m) {
      const double x = h->h_track[t][2 * m], y = h->h_track[t][2 * m + 1];
      if (!std::isfinite(x) || !std::isfinite(y)) continue;
      // the same expression the kernel uses for a particle: a point and a particle at the same place share a cell
      const int ix = std::min(std::max(static_cast<int>(std::floor((x - lo[0]) / cell)), 0), h->grid_nx - 1);
      const int iy = std::min(std::max(static_cast<int>(std::floor((y - lo[1]) / cell)), 0), h->grid_ny - 1);
      cell_of[m] = iy * h->grid_nx + ix;
      ++start[cell_of[m] + 1];
    }
    for (size_t c = 0; c < cells; ++c) start[c + 1] += start[c];
    const size_t kept = std::max<size_t>(static_cast<size_t>(start[cells]), 1);
    h->h_grid_index[t].assign(kept, 0);
    h->h_grid_x[t].assign(kept, 0.0);
    h->h_grid_y[t].assign(kept, 0.0);
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (size_t m = 0; m < M; ++m) {   // ascending m: ascending indices inside a cell
      if (cell_of[m] < 0) continue;
      const size_t j = static_cast<size_t>(fill[cell_of[m]]++);
      h->h_grid_index[t][j] = static_cast<int>(m);
      h->h_grid_x[t][j] = h->h_track[t][2 * m];
      h->h_grid_y[t][j] = h->h_track[t][2 * m + 1];
    }
  }
  // the 3 x 3 block round every cell as one run: its rows in order, each row a run of the cell-ordered arrays
  h->h_block_start.assign(3 * cells + 1, 0);
  h->h_block_index.clear();
  h->h_block_x.clear();
  h->h_block_y.clear();
  for (int t = 0; t < 3; ++t) {
    const std::vector<int>& start = h->h_grid_start[t];
    for (int iy = 0; iy < h->grid_ny; ++iy)
      for (int ix = 0; ix < h->grid_nx; ++ix) {
        h->h_block_start[t * cells + static_cast<size_t>(iy) * h->grid_nx + ix] = static_cast<int>(h->h_block_index.size());
        const int x_lo = std::max(ix - 1, 0), x_hi = std::min(ix + 1, h->grid_nx - 1);
        for (int cy = std::max(iy - 1, 0); cy <= std::min(iy + 1, h->grid_ny - 1); ++cy) {
          const int j0 = start[static_cast<size_t>(cy) * h->grid_nx + x_lo], j1 = start[static_cast<size_t>(cy) * h->grid_nx + x_hi + 1];
          h->h_block_index.insert(h->h_block_index.end(), h->h_grid_index[t].begin() + j0, h->h_grid_index[t].begin() + j1);
          h->h_block_x.insert(h->h_block_x.end(), h->h_grid_x[t].begin() + j0, h->h_grid_x[t].begin() + j1);
          h->h_block_y.insert(h->h_block_y.end(), h->h_grid_y[t].begin() + j0, h->h_grid_y[t].begin() + j1);
        }
      }
  }
  h->h_block_start[3 * cells] = static_cast<int>(h->h_block_index.size());
  if (h->h_block_index.empty()) {   // (nothing to allocate for: one unused entry)
    h->h_block_index.assign(1, 0);
    h->h_block_x.assign(1, 0.0);
    h->h_block_y.assign(1, 0.0);
  }
}

// from kGridParticles up the grid search is a launch of its own in front of the scoring (sixteen lanes per query); below,
// the scoring workgroup of a particle searches the grid itself, a wavefront per polyline
constexpr int kGridParticles = 4096;
bool use_grid(const acmpc_pf* h, int P) {
  return h->grid_nx > 0 && P >= kGridParticles && !h->no_grid;
}

GridBlocks blocks_of(const acmpc_pf* h) {
  if (h->no_blocks) return GridBlocks{nullptr, nullptr, nullptr, nullptr, 0};
  return GridBlocks{h->d_block_start, h->d_block_x, h->d_block_y, h->d_block_index, h->grid_nx * h->grid_ny};
}

// the grid search in front of a scoring launch (fills `a.given_*`) - or, below kGridParticles, inside it
hipError_t launch_nearest(acmpc_pf* h, ScoreArgs& a, int P, hipStream_t s) {
  a.own_grid_search = 0;
  if (h->grid_nx > 0 && !h->no_grid && P < kGridParticles) {
    a.own_grid_search = 1;
    for (int t = 0; t < 3; ++t) a.grid[t] = GridIndex{h->d_grid_start[t], h->d_grid_x[t], h->d_grid_y[t], h->d_grid_index[t]};
    a.geometry = GridGeometry{h->grid_x0, h->grid_y0, h->grid_cell, h->grid_nx, h->grid_ny};
    a.blocks = blocks_of(h);
    return hipSuccess;
  }
  if (!use_grid(h, P)) return hipSuccess;
  GridArgs g{};
  g.states = a.states;
  g.live = a.live;
  g.track[0] = a.centre;
  g.track[1] = a.left;
  g.track[2] = a.right;
  for (int t = 0; t < 3; ++t) g.grid[t] = GridIndex{h->d_grid_start[t], h->d_grid_x[t], h->d_grid_y[t], h->d_grid_index[t]};
  g.blocks = blocks_of(h);
  g.x0 = h->grid_x0;
  g.y0 = h->grid_y0;
  g.cell = h->grid_cell;
  g.nx = h->grid_nx;
  g.ny = h->grid_ny;
  g.index_out = h->d_near_index;
  g.d2_out = h->d_near_d2;
  hipLaunchKernelGGL(pf_nearest_kernel, dim3((3 * P + kQueriesPerWave - 1) / kQueriesPerWave), dim3(64), 0, s, g, P);
  a.given_index = h->d_near_index;
  a.given_d2 = h->d_near_d2;
  return hipGetLastError();
}

// the scoring launch behind launch_nearest: given nearest points -> a wavefront per 16 particles (4 where that would leave
// the chip short of waves); none given -> the workgroup kernels, which find them (PB = 1: through the grid)
void launch_score(const acmpc_pf* h, const ScoreArgs& a, int P, hipStream_t s) {
  if (a.given_index != nullptr && !h->workgroup_score) {
    ScoreArgs b = a;
    b.inv_left_m = 1.0f / static_cast<float>(a.left.m);
    b.inv_right_m = 1.0f / static_cast<float>(a.right.m);
    const int K = a.k_left + a.k_right;
    const bool one_trip = K <= kBlock;
    // the indices ahead of a nearest point: nearest + offset <= (m - 1) + max(k_left, k_right)
    const bool no_wrap = std::max(a.left.m, a.right.m) + K <= 65536 && K <= std::min(a.left.m, a.right.m);
    const int form = one_trip ? (no_wrap ? 0 : 1) : 2;
    // particles per wave: the launch as whole generations of resident waves (a handful of waves left over for a second
    // generation would double the kernel's time), at least 4 per wave, at most 32
    const int slots = std::max(h->given_wave_slots[form], 1);
    const int generations = (P + 32 * slots - 1) / (32 * slots);
    const int pw = std::min(std::max((P + slots * generations - 1) / (slots * generations), 4), 32);
    const dim3 grid((P + pw - 1) / pw);
    if (form == 0) hipLaunchKernelGGL((pf_score_given_kernel<true, true>), grid, dim3(64), 0, s, b, P, pw);
    else if (form == 1) hipLaunchKernelGGL((pf_score_given_kernel<true, false>), grid, dim3(64), 0, s, b, P, pw);
    else hipLaunchKernelGGL((pf_score_given_kernel<false, false>), grid, dim3(64), 0, s, b, P, pw);
  } else if (P >= kGridParticles) {
    constexpr int PB = 8;
    hipLaunchKernelGGL(pf_score_kernel<PB>, dim3((P + PB - 1) / PB), dim3(kBlock), 0, s, a, P);
  } else {
    hipLaunchKernelGGL(pf_score_kernel<1>, dim3(P), dim3(kBlock), 0, s, a, P);
  }
}

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
  if (h->grid_nx > 0) {
    for (int t = 0; t < 3; ++t) {
      PF_HIP(h, hipMalloc(&h->d_grid_start[t], h->h_grid_start[t].size() * sizeof(int)));
      PF_HIP(h, hipMemcpy(h->d_grid_start[t], h->h_grid_start[t].data(), h->h_grid_start[t].size() * sizeof(int),
                          hipMemcpyHostToDevice));
      const size_t kept = h->h_grid_index[t].size();
      PF_HIP(h, hipMalloc(&h->d_grid_index[t], kept * sizeof(int)));
      PF_HIP(h, hipMemcpy(h->d_grid_index[t], h->h_grid_index[t].data(), kept * sizeof(int), hipMemcpyHostToDevice));
      PF_HIP(h, hipMalloc(&h->d_grid_x[t], kept * sizeof(double)));
      PF_HIP(h, hipMemcpy(h->d_grid_x[t], h->h_grid_x[t].data(), kept * sizeof(double), hipMemcpyHostToDevice));
      PF_HIP(h, hipMalloc(&h->d_grid_y[t], kept * sizeof(double)));
      PF_HIP(h, hipMemcpy(h->d_grid_y[t], h->h_grid_y[t].data(), kept * sizeof(double), hipMemcpyHostToDevice));
    }
    PF_HIP(h, hipMalloc(&h->d_block_start, h->h_block_start.size() * sizeof(int)));
    PF_HIP(h, hipMemcpy(h->d_block_start, h->h_block_start.data(), h->h_block_start.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t in_blocks = h->h_block_index.size();
    PF_HIP(h, hipMalloc(&h->d_block_index, in_blocks * sizeof(int)));
    PF_HIP(h, hipMemcpy(h->d_block_index, h->h_block_index.data(), in_blocks * sizeof(int), hipMemcpyHostToDevice));
    PF_HIP(h, hipMalloc(&h->d_block_x, in_blocks * sizeof(double)));
    PF_HIP(h, hipMemcpy(h->d_block_x, h->h_block_x.data(), in_blocks * sizeof(double), hipMemcpyHostToDevice));
    PF_HIP(h, hipMalloc(&h->d_block_y, in_blocks * sizeof(double)));
    PF_HIP(h, hipMemcpy(h->d_block_y, h->h_block_y.data(), in_blocks * sizeof(double), hipMemcpyHostToDevice));
    PF_HIP(h, hipMalloc(&h->d_near_index, P * 3 * sizeof(int32_t)));
    PF_HIP(h, hipMalloc(&h->d_near_d2, P * 3 * sizeof(double)));
  }
  PF_HIP(h, hipMalloc(&h->d_states, (P * 3 + K * 2) * sizeof(float)));
  h->d_obs = nullptr;  // placed behind the states of each call
  PF_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_up), (P * 3 + K * 2) * sizeof(float), hipHostMallocDefault));
  const size_t down = P * (4 * sizeof(double) + 3 * sizeof(int32_t) + 1);
  PF_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_down), down, hipHostMallocDefault));
  PF_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->d_down), down));
  PF_HIP(h, hipMalloc(&h->d_aux, P * 2 * sizeof(float)));
  PF_HIP(h, hipMalloc(&h->d_out, 8 * sizeof(double)));
  {
    // how many waves of the wave-per-particles scoring kernel the device holds at once (launch_score sizes its waves by it)
    int device = 0, units = 0, per_unit[3] = {0, 0, 0};
    PF_HIP(h, hipGetDevice(&device));
    PF_HIP(h, hipDeviceGetAttribute(&units, hipDeviceAttributeMultiprocessorCount, device));
    PF_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit[0], pf_score_given_kernel<true, true>, 64, 0));
    PF_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit[1], pf_score_given_kernel<true, false>, 64, 0));
    PF_HIP(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_unit[2], pf_score_given_kernel<false, false>, 64, 0));
    for (int f = 0; f < 3; ++f) h->given_wave_slots[f] = units * per_unit[f];
  }
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
  {
    const char* value = std::getenv("ACMPC_PF_NO_GRID");
    h->no_grid = value != nullptr && value[0] != '\0' && !(value[0] == '0' && value[1] == '\0');
    value = std::getenv("ACMPC_PF_WORKGROUP_SCORE");
    h->workgroup_score = value != nullptr && value[0] != '\0' && !(value[0] == '0' && value[1] == '\0');
    value = std::getenv("ACMPC_PF_NARROW_FILTER");
    h->narrow_filter = value != nullptr && value[0] != '\0' && !(value[0] == '0' && value[1] == '\0');
    value = std::getenv("ACMPC_PF_NO_BLOCKS");
    h->no_blocks = value != nullptr && value[0] != '\0' && !(value[0] == '0' && value[1] == '\0');
  }
  h->h_track[0].assign(centre, centre + 2 * static_cast<size_t>(m_centre));
  h->h_track[1].assign(left, left + 2 * static_cast<size_t>(m_left));
  h->h_track[2].assign(right, right + 2 * static_cast<size_t>(m_right));
  build_grid(h);
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
    for (int t = 0; t < 3; ++t) {
      (void)hipFree(h->d_track[t]);
      (void)hipFree(h->d_grid_start[t]);
      (void)hipFree(h->d_grid_index[t]);
      (void)hipFree(h->d_grid_x[t]);
      (void)hipFree(h->d_grid_y[t]);
    }
    (void)hipFree(h->d_block_start);
    (void)hipFree(h->d_block_index);
    (void)hipFree(h->d_block_x);
    (void)hipFree(h->d_block_y);
    (void)hipFree(h->d_near_index);
    (void)hipFree(h->d_near_d2);
    (void)hipFree(h->d_states);
    (void)hipFree(h->d_down);
    if (h->h_up != nullptr) (void)hipHostFree(h->h_up);
    if (h->h_down != nullptr) (void)hipHostFree(h->h_down);
    (void)hipFree(h->d_aux);
    (void)hipFree(h->d_out);
    for (int b = 0; b < 2; ++b) {
      (void)hipFree(h->f_states[b]);
      (void)hipFree(h->f_scores[b]);
    }
    (void)hipFree(h->f_obs);
    (void)hipFree(h->f_cdf);
    (void)hipFree(h->f_kept);
    (void)hipFree(h->f_counts);
    (void)hipFree(h->f_wide);
    (void)hipFree(h->f_partial);
    if (h->h_result != nullptr) (void)hipHostFree(h->h_result);
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
  const int K = k_left + k_right;
  const size_t up_floats = static_cast<size_t>(P) * 3 + static_cast<size_t>(K) * 2;
  float* up = reinterpret_cast<float*>(h->h_up);
  std::memcpy(up, states, static_cast<size_t>(P) * 3 * sizeof(float));
  if (k_left > 0) std::memcpy(up + 3 * static_cast<size_t>(P), obs_left, static_cast<size_t>(k_left) * 2 * sizeof(float));
  if (k_right > 0)
    std::memcpy(up + 3 * static_cast<size_t>(P) + 2 * static_cast<size_t>(k_left), obs_right,
                static_cast<size_t>(k_right) * 2 * sizeof(float));
  PF_HIP(h, hipMemcpyAsync(h->d_states, up, up_floats * sizeof(float), hipMemcpyHostToDevice, s));
  const size_t pd = static_cast<size_t>(P) * sizeof(double);
  ScoreArgs a{};
  a.states = h->d_states;
  a.obs = h->d_states + 3 * static_cast<size_t>(P);
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
  // the reference's particle counts (hundreds): the scoring kernel writes its results straight into the page-locked block
  // the host reads them from (posted writes over the host link; page-locked memory is device-addressable) - no copy
  // packet behind the kernel.  Large counts keep the copy: megabytes of 8-byte pieces are better moved as one DMA.
  const bool in_place = P < 4096;
  unsigned char* out = in_place ? h->h_down : h->d_down;
  a.minimum_offset = reinterpret_cast<double*>(out);
  a.heading_offset = a.minimum_offset + P;
  a.error = a.heading_offset + P;
  a.score = a.error + P;
  a.track_indices = reinterpret_cast<int32_t*>(out + 4 * pd);
  a.valid = reinterpret_cast<uint8_t*>(out + 4 * pd + static_cast<size_t>(P) * 3 * sizeof(int32_t));
  (void)hipGetLastError();  // a stale error of an earlier call must not be read as this launch's
  PF_HIP(h, launch_nearest(h, a, P, s));
  launch_score(h, a, P, s);
  PF_HIP(h, hipGetLastError());
  const size_t down = 4 * pd + static_cast<size_t>(P) * (3 * sizeof(int32_t) + 1);
  if (!in_place) PF_HIP(h, hipMemcpyAsync(h->h_down, h->d_down, down, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipStreamSynchronize(s));
  const unsigned char* dn = h->h_down;
  std::memcpy(minimum_offset, dn, pd);
  std::memcpy(heading_offset, dn + pd, pd);
  std::memcpy(observation_error, dn + 2 * pd, pd);
  std::memcpy(score, dn + 3 * pd, pd);
  std::memcpy(track_indices, dn + 4 * pd, static_cast<size_t>(P) * 3 * sizeof(int32_t));
  std::memcpy(valid, dn + 4 * pd + static_cast<size_t>(P) * 3 * sizeof(int32_t), static_cast<size_t>(P));
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
  if (P < 4096) {
    // the reference's particle counts: every particle is read and written once by its own lane, so the kernel works IN
    // the page-locked block (20 bytes per particle of the 45 it holds) - four copies from and to pageable memory, a
    // synchronous staged copy of ~15 us each whatever its size, were most of this call
    float* block = reinterpret_cast<float*>(h->h_down);
    std::memcpy(block, states, 3 * pf);
    std::memcpy(block + 3 * static_cast<size_t>(P), delta, pf);
    std::memcpy(block + 4 * static_cast<size_t>(P), velocity, pf);
    (void)hipGetLastError();
    hipLaunchKernelGGL(pf_advance_kernel, dim3((P + 255) / 256), dim3(256), 0, s, block, block + 3 * static_cast<size_t>(P),
                       block + 4 * static_cast<size_t>(P), P, static_cast<float>(h->prm.wheelbase), static_cast<float>(dt));
    PF_HIP(h, hipGetLastError());
    PF_HIP(h, hipStreamSynchronize(s));
    std::memcpy(states, block, 3 * pf);
    return ACMPC_OK;
  }
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
  double* d_res = h->d_out;
  (void)hipGetLastError();  // a stale error of an earlier call must not be read as this launch's
  hipLaunchKernelGGL(pf_estimate_kernel, dim3(1), dim3(kBlock), 0, s, h->d_states, h->d_aux, P, d_res,
                     static_cast<const int*>(nullptr));
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

namespace {

// the tiled resampling and estimate (pf_resample_tiles_kernel ...) for capacities one workgroup is too slow for; the plan
// kernel holds a tile per thread, which bounds them at 2^20 particles.  ACMPC_PF_NARROW_FILTER=1: one workgroup always (A/B)
bool wide_filter(const acmpc_pf* h) {
  return h->prm.max_particles >= kWideFilter && h->prm.max_particles <= kScanBlock * kScanBlock && !h->narrow_filter;
}

int pf_ensure_filter(acmpc_pf* h) {
  if (h->filter_ready) return ACMPC_OK;
  const int rc = pf_ensure_device(h);
  if (rc != ACMPC_OK) return rc;
  const size_t P = h->prm.max_particles, K = h->prm.max_observation_points;
  for (int b = 0; b < 2; ++b) {
    PF_HIP(h, hipMalloc(&h->f_states[b], P * 3 * sizeof(float)));
    PF_HIP(h, hipMalloc(&h->f_scores[b], P * sizeof(float)));
  }
  PF_HIP(h, hipMalloc(&h->f_obs, K * 2 * sizeof(float)));
  PF_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->f_cdf), P * sizeof(unsigned long long)));
  PF_HIP(h, hipMalloc(&h->f_kept, P * sizeof(int)));
  PF_HIP(h, hipMalloc(&h->f_counts, 4 * sizeof(int)));
  if (wide_filter(h)) {
    const size_t tiles = (P + kScanBlock - 1) / kScanBlock, sums = (P + kEstimateTile - 1) / kEstimateTile;
    PF_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->f_wide), (2 * tiles + 6) * sizeof(unsigned long long)));
    PF_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->f_partial), sums * 10 * sizeof(double)));
  }
  PF_HIP(h, hipMemset(h->f_counts, 0, 4 * sizeof(int)));
  PF_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_result), 8 * sizeof(double) + 4 * sizeof(int), hipHostMallocDefault));
  PF_HIP(h, hipStreamSynchronize(nullptr));
  h->filter_ready = true;
  return ACMPC_OK;
}

FilterArgs filter_args(acmpc_pf* h, const acmpc_pf_resample* rs) {
  FilterArgs f{};
  const size_t P = h->prm.max_particles;
  f.states_in = h->f_states[h->f_cur];
  f.scores_in = h->f_scores[h->f_cur];
  f.score = reinterpret_cast<const double*>(h->d_down) + 3 * P;
  f.valid = reinterpret_cast<const uint8_t*>(h->d_down + 4 * P * sizeof(double) + P * 3 * sizeof(int32_t));
  f.states_out = h->f_states[1 - h->f_cur];
  f.scores_out = h->f_scores[1 - h->f_cur];
  f.cdf = h->f_cdf;
  f.kept = h->f_kept;
  f.counts = h->f_counts;
  f.centre = h->d_track[0];
  f.m_centre = static_cast<int>(h->h_track[0].size() / 2);
  f.capacity = h->prm.max_particles;
  if (rs != nullptr) {
    f.n_desired = rs->n_desired;
    f.minimum_particles = rs->minimum_particles;
    f.sigma_x = rs->sigma_x;
    f.sigma_y = rs->sigma_y;
    f.sigma_yaw = rs->sigma_yaw;
    f.seed_lo = static_cast<uint32_t>(rs->seed);
    f.seed_hi = static_cast<uint32_t>(rs->seed >> 32);
    f.counter = rs->counter;
  }
  return f;
}

}  // namespace

extern "C" {

int acmpc_pf_filter_reset(acmpc_pf* h, int32_t n) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (n < 1 || n > h->prm.max_particles) return pf_fail(h, ACMPC_ECAPACITY, "particle count out of range");
  const int rc = pf_ensure_filter(h);
  if (rc != ACMPC_OK) return rc;
  // the resampling kernel's reset branch with nothing valid: n_live = 0, minimum = 1, capacity = n
  hipStream_t s = h->stream;
  const int zero[4] = {0, 0, 0, 0};
  PF_HIP(h, hipMemcpyAsync(h->f_counts, zero, sizeof zero, hipMemcpyHostToDevice, s));
  FilterArgs f = filter_args(h, nullptr);
  f.capacity = n;
  f.minimum_particles = 1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(pf_resample_kernel, dim3(1), dim3(kScanBlock), 0, s, f);
  PF_HIP(h, hipGetLastError());
  PF_HIP(h, hipStreamSynchronize(s));
  h->f_cur = 1 - h->f_cur;
  return ACMPC_OK;
}

int acmpc_pf_filter_set(acmpc_pf* h, const float* states, const float* scores, int32_t n) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (states == nullptr || scores == nullptr) return pf_fail(h, ACMPC_EINVAL, "null argument");
  if (n < 1 || n > h->prm.max_particles) return pf_fail(h, ACMPC_ECAPACITY, "particle count out of range");
  const int rc = pf_ensure_filter(h);
  if (rc != ACMPC_OK) return rc;
  hipStream_t s = h->stream;
  const int counts[4] = {n, 0, 0, 0};
  PF_HIP(h, hipMemcpyAsync(h->f_states[h->f_cur], states, static_cast<size_t>(n) * 3 * sizeof(float), hipMemcpyHostToDevice, s));
  PF_HIP(h, hipMemcpyAsync(h->f_scores[h->f_cur], scores, static_cast<size_t>(n) * sizeof(float), hipMemcpyHostToDevice, s));
  PF_HIP(h, hipMemcpyAsync(h->f_counts, counts, sizeof counts, hipMemcpyHostToDevice, s));
  PF_HIP(h, hipStreamSynchronize(s));
  return ACMPC_OK;
}

int acmpc_pf_filter_get(acmpc_pf* h, float* states, float* scores, int32_t capacity, int32_t* n) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (states == nullptr || scores == nullptr || n == nullptr) return pf_fail(h, ACMPC_EINVAL, "null argument");
  if (!h->filter_ready) return pf_fail(h, ACMPC_ESTATE, "no particles on the device yet");
  hipStream_t s = h->stream;
  int counts[4];
  PF_HIP(h, hipMemcpyAsync(counts, h->f_counts, sizeof counts, hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipStreamSynchronize(s));
  if (counts[0] > capacity) return pf_fail(h, ACMPC_ECAPACITY, "output buffers too small");
  PF_HIP(h, hipMemcpyAsync(states, h->f_states[h->f_cur], static_cast<size_t>(counts[0]) * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipMemcpyAsync(scores, h->f_scores[h->f_cur], static_cast<size_t>(counts[0]) * sizeof(float), hipMemcpyDeviceToHost, s));
  PF_HIP(h, hipStreamSynchronize(s));
  *n = counts[0];
  return ACMPC_OK;
}

int acmpc_pf_filter_step(acmpc_pf* h, double tyre_angle, double velocity, double dt, double sigma_yaw,
                         double sigma_velocity, uint64_t seed, uint32_t counter) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (!h->filter_ready) return pf_fail(h, ACMPC_ESTATE, "no particles on the device yet");
  const int P = h->prm.max_particles;
  (void)hipGetLastError();
  hipLaunchKernelGGL(pf_step_kernel, dim3((P + 255) / 256), dim3(256), 0, h->stream, h->f_states[h->f_cur], h->f_counts,
                     static_cast<float>(tyre_angle), static_cast<float>(velocity), static_cast<float>(sigma_yaw),
                     static_cast<float>(sigma_velocity), static_cast<float>(h->prm.wheelbase), static_cast<float>(dt),
                     static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), counter);
  PF_HIP(h, hipGetLastError());
  return ACMPC_OK;
}

int acmpc_pf_filter_update(acmpc_pf* h, const float* obs_left, int32_t k_left, const float* obs_right,
                           int32_t k_right, const acmpc_pf_resample* rs, double* result) {
  if (h == nullptr) return ACMPC_EINVAL;
  if (rs == nullptr || result == nullptr) return pf_fail(h, ACMPC_EINVAL, "null argument");
  if (rs->struct_size != sizeof(acmpc_pf_resample)) return pf_fail(h, ACMPC_EINVAL, "acmpc_pf_resample size mismatch");
  if (k_left < 0 || k_right < 0 || k_left + k_right < 1) return pf_fail(h, ACMPC_EINVAL, "empty observation");
  if ((k_left > 0 && obs_left == nullptr) || (k_right > 0 && obs_right == nullptr))
    return pf_fail(h, ACMPC_EINVAL, "null observation");
  if (k_left + k_right > h->prm.max_observation_points) return pf_fail(h, ACMPC_ECAPACITY, "too many observation points");
  if (!h->filter_ready) return pf_fail(h, ACMPC_ESTATE, "no particles on the device yet");
  hipStream_t s = h->stream;
  const size_t Pmax = h->prm.max_particles;
  // observation up (one pinned block), everything else is already there
  float* up = reinterpret_cast<float*>(h->h_up);
  if (k_left > 0) std::memcpy(up, obs_left, static_cast<size_t>(k_left) * 2 * sizeof(float));
  if (k_right > 0) std::memcpy(up + 2 * static_cast<size_t>(k_left), obs_right, static_cast<size_t>(k_right) * 2 * sizeof(float));
  PF_HIP(h, hipMemcpyAsync(h->f_obs, up, static_cast<size_t>(k_left + k_right) * 2 * sizeof(float), hipMemcpyHostToDevice, s));
  ScoreArgs a{};
  a.states = h->f_states[h->f_cur];
  a.obs = h->f_obs;
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
  const size_t pd = Pmax * sizeof(double);
  a.minimum_offset = reinterpret_cast<double*>(h->d_down);
  a.heading_offset = a.minimum_offset + Pmax;
  a.error = a.heading_offset + Pmax;
  a.score = a.error + Pmax;
  a.track_indices = reinterpret_cast<int32_t*>(h->d_down + 4 * pd);
  a.valid = reinterpret_cast<uint8_t*>(h->d_down + 4 * pd + Pmax * 3 * sizeof(int32_t));
  a.live = h->f_counts;   // the particle count is on the device
  a.publish = h->f_scores[h->f_cur];   // (the scoring kernels write the float32 scores themselves: no launch for that)
  (void)hipGetLastError();
  {
    const int P = h->prm.max_particles;   // workgroups beyond the live count return at once
    PF_HIP(h, launch_nearest(h, a, P, s));
    launch_score(h, a, P, s);
  }
  PF_HIP(h, hipGetLastError());
  const FilterArgs f = filter_args(h, rs);
  const bool wide = wide_filter(h);
  if (wide) {
    const int tiles = static_cast<int>((Pmax + kScanBlock - 1) / kScanBlock);
    const WideScratch w{h->f_wide, h->f_wide + tiles, h->f_wide + 2 * tiles};
    hipLaunchKernelGGL(pf_resample_tiles_kernel, dim3(tiles), dim3(kScanBlock), 0, s, f, w);
    hipLaunchKernelGGL(pf_resample_plan_kernel, dim3(1), dim3(kScanBlock), 0, s, f, w, tiles);
    hipLaunchKernelGGL(pf_resample_scatter_kernel, dim3(tiles), dim3(kScanBlock), 0, s, f, w);
    hipLaunchKernelGGL(pf_resample_emit_kernel, dim3(static_cast<unsigned>((Pmax + 255) / 256)), dim3(256), 0, s, f, w);
  } else {
    hipLaunchKernelGGL(pf_resample_kernel, dim3(1), dim3(kScanBlock), 0, s, f);
  }
  PF_HIP(h, hipGetLastError());
  h->f_cur = 1 - h->f_cur;
  double* res = h->h_result;
  // the estimate and the counts are written by the last kernel straight into the page-locked result block (posted writes
  // over the host link): no copy packets behind the update
  if (wide) {
    const int sums = static_cast<int>((Pmax + kEstimateTile - 1) / kEstimateTile);
    double* spread = h->f_partial + static_cast<size_t>(sums) * 8;
    hipLaunchKernelGGL(pf_estimate_partial_kernel, dim3(sums), dim3(kBlock), 0, s, h->f_states[h->f_cur], h->f_scores[h->f_cur],
                       h->f_counts, h->f_partial);
    hipLaunchKernelGGL(pf_estimate_spread_kernel, dim3(sums), dim3(kBlock), 0, s, h->f_states[h->f_cur], h->f_counts,
                       h->f_partial, spread);
    hipLaunchKernelGGL(pf_estimate_final_kernel, dim3(1), dim3(kBlock), 0, s, h->f_counts, h->f_partial, spread, res,
                       reinterpret_cast<int*>(res + 8));
  } else {
    hipLaunchKernelGGL(pf_estimate_kernel, dim3(1), dim3(kBlock), 0, s, h->f_states[h->f_cur], h->f_scores[h->f_cur], -1,
                       res, h->f_counts, reinterpret_cast<int*>(res + 8));
  }
  PF_HIP(h, hipGetLastError());
  PF_HIP(h, hipStreamSynchronize(s));
  const int* counts = reinterpret_cast<const int*>(res + 8);
  for (int i = 0; i < 5; ++i) result[i] = res[i];
  result[5] = counts[0];
  result[6] = counts[1];
  result[7] = counts[2];
  return ACMPC_OK;
}

}  // extern "C"
