// Host-callable launchers of the gfx950 kernels (definitions in acmpc_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acmpc_device.h"

namespace acmpc {

struct RolloutArgs {
  const float* U;         // control-sample matrix, layout per `layout`
  const float* x0;        // [P][3]
  const float* coef;      // [P][n][stride] packed per-step table
  const float* nn_frames; // mode T, exhaustive search: [P][verified_frame_floats(n)] frames of the verified window search, or nullptr
  float* costs;           // [P][N] or nullptr
  int64_t* partial_keys;  // [P][blocks_per_problem]
  int* partial_feas;      // [P][blocks_per_problem]
  int P, N, n;
  int64_t index_offset;   // global index of local candidate 0
  Weights w;
  // mode T, set by the launcher for a launch that is ONE generation of waves (every workgroup resident from the start):
  // a wave lowers its own issue priority as it moves through the horizon, so that the waves of a SIMD finish together
  // instead of oldest first
  int even_progress = 0;
  // diagnostic (acmpc_set_option ACMPC_START_CLOCKS): when not null, every workgroup leaves the 100 MHz wall clock of its
  // first instruction at [p * blocks_per_problem + block] - what shows a launch whose workgroups did not all start together
  unsigned long long* start_clock = nullptr;
};

struct FinalizeArgs {
  const float* U;
  const float* x0;
  const float* coef;
  const int64_t* partial_keys;  // used when keys_in == nullptr
  const int* partial_feas;
  const int64_t* keys_in;       // [P] global keys (after an all-reduce) or nullptr
  int64_t* keys_out;            // [P] or nullptr
  float* records;               // [P][record_floats] or nullptr
  // regenerate == true: the winner's controls are re-drawn from its global index (counter-based sampler) instead
  // of being loaded from U, so EVERY rank can write the full record after one all-reduce(MIN) of the keys
  bool regenerate;
  // controls_only == true: write header + u block only (cost taken from the key, violation 0, no re-roll, no x
  // block) - what the next round of an optimisation needs from the previous one
  bool controls_only;
  const float* centre;
  const float* u_ref;
  const float* u_extra;         // [P][n][2] or nullptr: candidate 2 (see SampleArgs)
  int centre_stride;
  SampleSpec spec;
  int blocks_per_problem;
  int P, N, n;
  int64_t index_offset;
  Weights w;
};

struct SoftminArgs {
  const float* costs;     // [P][N]
  const int64_t* keys;    // [P]
  const float* U;
  double* partial;        // [P][chunks][2n + 1] workspace
  float* mean;            // [P][n][2]
  double* weight_sum;     // [P] or nullptr
  int chunks;
  int P, N, n;
  float lambda;
};

constexpr int kSampleKnots = kKnots;

struct SampleArgs {
  const float* centre;     // [P] x centre_stride floats, first 2n of each = (v, kappa) per step
  const float* u_ref;      // [P][n][2] or nullptr: becomes candidate 1
  const float* u_extra;    // [P][n][2] or nullptr: becomes candidate 2 (the LQ plan of csrc/acmpc_lq.h: the unconstrained
                           // optimum of the QP, rolled forward with its feedback and clipped into the input box)
  float* U;                // out, layout per `layout`
  int centre_stride;
  int P, N, n;
  int64_t index_offset;    // global index of local candidate 0 (the counter of the generator)
  SampleSpec spec;
  // rollout_sampled_kernel only - the centre taken straight from the PREVIOUS round's launch instead of `centre`: the
  // argmin over that launch's partial keys names a workgroup, and the head of its trace is the winner's controls
  const int64_t* prev_keys;  // [P][prev_blocks] or nullptr
  const float* prev_trace;   // [P][prev_blocks][prev_pitch]
  int prev_blocks;           // <= kChainBlocks
  int prev_pitch;
};
constexpr int kTileRowsMaxSteps = 80;   // longest horizon of rollout_tile_rows_kernel (2 VGPRs per step)
constexpr int kChainBlocks = 256;   // four keys per lane

// The A/B switches of the tests and the tools.  They are read from the environment ONCE, by acmpc_create (ACMPC_* variables,
// tools/README.md), or set through acmpc_set_option, and travel with the handle: nothing on a launch path calls getenv, and
// an environment variable that appears later changes nothing.  The defaults are the shipped forms.
struct LaunchOptions {
  int shape_block = 0, shape_cpt = 0;   // ACMPC_SHAPE="<block>,<cpt>": the rollout's launch shape
  int temporal_pack = 0;                // ACMPC_T_PACK: 1 = plain float32 states, 2 = packed pairs (0: the default, plain)
  bool no_tile = false;                 // ACMPC_NO_TILE
  int tile_rows = -1;                   // ACMPC_TILE_ROWS: 4 = the rows-in-registers tile kernel, 0 = not (-1: by size)
  int tile_table = 0;                   // ACMPC_TILE_TABLE: 1 = lds, 2 = scalar (0: by shape)
  bool no_trio_rounds = false, no_quad_rounds = false, no_pair_rounds = false;
  int solo_registers = -1, solo_split = -1;   // ACMPC_SOLO_REGISTERS / ACMPC_SOLO_SPLIT: 0 / 1 (-1: by size / split)
  bool no_group_finalize = false;             // ACMPC_NO_GROUP_FINALIZE: a wavefront per problem in the batched finalize at any problem count
  bool finalize_waves = false;                // ACMPC_FINALIZE_WAVES: the many-problem finalize on a wavefront per problem, four per workgroup
};

struct LaunchShape {
  int block;              // threads per workgroup
  int cpt;                // candidates per thread
  int blocks_per_problem;
  bool tile;              // candidate-major LDS-tile kernel
  int tile_waves;         // tile kernel with the rows in registers: waves per workgroup sharing one LDS tile (0: not used)
  int pack;               // mode T, cpt >= 2: candidates per arithmetic state (2 = v_pk_* pairs, 1 = plain float32)
  int tile_table = 0;     // rows-in-registers tile kernel: 1 = table rows from LDS, 2 = by scalar loads (0: by shape)
};

// Picks workgroup size / candidates per thread for (P, N, layout); pure function, also used to size workspaces.
LaunchShape choose_shape(int P, int N, int layout, int mode, int n, const LaunchOptions& opt = LaunchOptions());
int max_blocks_per_problem(int N);
size_t tile_lds_bytes(int mode, int n);

// `start`/`stop` (both or neither) are attached to the dispatch itself (hipExtLaunchKernel): the kernel's own begin
// and end timestamps, with no marker packets added to the stream.
hipError_t launch_rollout(int mode, int layout, const LaunchShape& shape, const RolloutArgs& args, hipStream_t s,
                          hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// (defined in acmpc_kernels_temporal.hip, the translation unit built without the SLP vectoriser)
hipError_t launch_rollout_temporal_plain(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s,
                                         hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_tile_rows_plain(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s,
                                          hipEvent_t start, hipEvent_t stop);
hipError_t launch_finalize(int mode, int layout, const FinalizeArgs& args, hipStream_t s,
                           const LaunchOptions& opt = LaunchOptions());
// rollout + finalize in ONE launch for the batched solve (mode S, step-major, 256-thread shapes): the last workgroup of
// every problem writes its record (`fin`: a FinalizeArgs as launch_finalize takes it, partial keys = the rollout's own;
// `tickets` [P][kTicketGroups + 1] counters, kTicketStride ints apart, zero before and after)
bool chained_rollout_fits(int mode, int layout, const LaunchShape& shape, int P, const FinalizeArgs& fin, int fin_layout);
hipError_t launch_rollout_chained(int layout, const LaunchShape& shape, const RolloutArgs& args, const FinalizeArgs& fin,
                                  int fin_layout, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
bool tailed_rollout_fits(int mode, int layout, const LaunchShape& shape, int n);
hipError_t launch_rollout_tailed(int layout, const LaunchShape& shape, const RolloutArgs& args, const FinalizeArgs& fin,
                                 int* tickets, hipStream_t s, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
hipError_t launch_sample(int layout, const SampleArgs& args, hipStream_t s);
// Optional tail of the fused launch: the workgroup that finishes a problem LAST (two levels of ticket counters per
// problem) also runs the finalize for it, in the same launch - argmin over the partial keys, record written.  With
// `trace` the record is copied out of what the winning workgroup left there (every workgroup writes the controls,
// states, violation and cost of its best candidate: [P][workgroups][trace_pitch] floats); without, the winner is
// re-drawn from its index and rolled again.  `tickets` [P][kTicketGroups + 1] counters, kTicketStride ints apart, must be zero before the
// launch and are left zero by it.
constexpr int kTicketGroups = 8;
// every counter on a 256-byte line of its own: device-scope atomics on ONE line serialise at ~6-13 ns each whichever
// word of it they hit (1 024 workgroups on 33 adjacent counters queued for 6 us), lines of their own do not
constexpr int kTicketStride = 64;      // ints between two counters
constexpr int kTicketGroupsMax = 32;   // rollout_solo_kernel's launches of more than 256 workgroups; `tickets` is sized for it
constexpr int kSoloBlocks = 1024;      // workgroups per launch of rollout_solo_kernel (16 partial keys per lane of the last one)
struct FusedFinalize {
  int* tickets;        // nullptr: no fused finalize
  float* records;      // [P][record_floats]
  bool controls_only;  // see FinalizeArgs
  float* trace;        // nullptr: re-draw and re-roll the winner
  int trace_pitch;     // floats per workgroup in `trace` (>= trace_floats(n))
  // completion flag (pinned host memory) or nullptr: `done_value` is stored there, at system scope, AFTER problem 0's
  // record - a host that polls it has the record without waiting for the launch's completion signal
  unsigned* done;
  unsigned done_value;
  // rollout_solo_kernel only
  int64_t* keys_out = nullptr;   // [P] winners' keys, or nullptr
  int ticket_groups = 0;         // power of two <= kTicketGroupsMax (set by the launcher)
};

// sample + rollout + cost (+ finalize) fused: candidates are drawn inside the rollout kernel and never touch memory
// (the closed-loop solve, where every launch is ~10 us of latency-bound work)
hipError_t launch_rollout_sampled(int mode, const RolloutArgs& rollout, const SampleArgs& sample,
                                  const FusedFinalize& fused, hipStream_t s, hipEvent_t start = nullptr,
                                  hipEvent_t stop = nullptr, const LaunchOptions& opt = LaunchOptions());
// acmpc_solve_device in ONE launch (mode S, at most kSoloBlocks workgroups of 64 candidates): rollout, argmin and the
// winner's record assembled from the winning workgroup's state trace - `fused.trace` [P][workgroups][trace_pitch >=
// solo_trace_floats(n)], `fused.tickets` [P][ticket_groups + 1] counters (kTicketStride ints apart) zero before and after.  `fused.records` may be
// null (keys only).
bool solo_fits(int P, int N, int n, int layout, const LaunchOptions& opt = LaunchOptions());
int solo_trace_floats(int n);
hipError_t launch_rollout_solo(int layout, const RolloutArgs& args, const FusedFinalize& fused, hipStream_t s,
                               hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                               const LaunchOptions& opt = LaunchOptions());
int trace_floats(int n);
bool trio_frames_fit(int n);   // mode T three-wave round: do the verified search's frames fit its LDS at this horizon?
bool traced_finalize_fits(int mode, int n);
// whether rollout + fused finalize fit one workgroup's 64 KB of LDS (mode T at the longest horizons does not)
bool fused_finalize_fits(int mode, int n);
int softmin_chunks(int N);
hipError_t launch_softmin(int layout, const SoftminArgs& args, hipStream_t s);

}  // namespace acmpc
