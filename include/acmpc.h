/*
 * acmpc.h - C ABI of the MI355X (gfx950) rollout-and-cost engine for the ac-mpc controller.
 *
 * This is the drop-in boundary of SURVEY.md section 8(b).  The reference is pure Python, so there is no
 * FFI to bind to; each entry point names the reference interface whose work it takes over
 * (paths relative to /root/reference/).  Host code stays Python and calls these through ctypes
 * (`ac-mpc_amd/acmpc_amd/_capi.py`; reference-side stub in INTEGRATION.md).
 *
 * Conventions
 *   - every function returns 0 on success or a negative ACMPC_E* code; no exception crosses the ABI;
 *     `acmpc_last_error()` gives the message of the last failure on that handle;
 *   - a handle is not thread-safe; buffers passed in are only read for the duration of the call;
 *   - a handle lives on ONE device (`acmpc_params.device`, or the thread's current device when -1): the calling
 *     thread's current HIP device must be that device on every call (one process per GPU does this once);
 *   - `acmpc_create`, `acmpc_set_paths` and `acmpc_get_coefficients` do no device work: the reference builds
 *     its MPC objects in the parent process and then forks the ControlProcess
 *     (src/acmpc/control/controller.py:94-100,293-297), so the HIP context is created lazily on the first call
 *     that needs the GPU, in whichever process makes it;
 *   - "n" is the number of control steps = horizon - 1 = len(ReferencePath)
 *     (src/acmpc/control/spatial_mpc.py:133-134), "N" the number of candidate control sequences per problem,
 *     "P" the number of independent problems (poses) evaluated by one launch;
 *   - a candidate is n pairs (v, kappa): the decision variables u of the reference QP
 *     (src/acmpc/control/solvers/control.py:26-28); steering angle = atan(kappa * wheelbase)
 *     (spatial_mpc.py:195-196).
 *
 * Rollout arithmetic is float32 in one fixed operation order with no implicit FMA contraction: mode S uses no
 * fused multiply-add at all, mode T uses exactly the ones its specification names (the
 * oracle calls fmaf / emulates it exactly), so results are bit-identical to oracle/acmpc_oracle.{py,c}.
 * The per-tick prologue (acmpc_control_tick) is float64; its tolerances are stated in DESIGN.md section 2.
 */
#ifndef ACMPC_H
#define ACMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACMPC_OK 0
#define ACMPC_EINVAL (-1)    /* bad argument (null pointer, size out of range, unknown mode/layout)      */
#define ACMPC_EHIP (-2)      /* a HIP runtime call failed; message carries hipGetErrorString             */
#define ACMPC_ENODEVICE (-3) /* no usable GPU / HIP runtime: the product path never falls back to the CPU */
#define ACMPC_ECAPACITY (-4) /* P, N or n exceeds what the handle was created for                         */
#define ACMPC_ESTATE (-5)    /* call order violated (e.g. solve before set_paths)                          */

/* Rollout model. */
#define ACMPC_MODE_SPATIAL 0  /* x_{i+1} = A_i x_i + B_i (u_i - u_ref_i) + f_i, x = (e_y, e_psi, t):
                                 dynamics.py:65-103 + control.py:26-45 ("mode S")                        */
#define ACMPC_MODE_TEMPORAL 1 /* Cartesian kinematic Euler step (localisation/localiser.py:66-95) with
                                 nearest-waypoint Frenet projection (localiser.py:282-289, dynamics.py:23-40)
                                 ("mode T")                                                               */

/* Memory layout of the control-sample matrix U. */
#define ACMPC_LAYOUT_CANDIDATE_MAJOR 0 /* U[P][N][n][2]  - what NumPy host code naturally holds          */
#define ACMPC_LAYOUT_STEP_MAJOR 1      /* U[P][n][2][N]  - what an on-device sampler writes; lane-coalesced */

/* Width in floats of one row of the packed per-step tables returned by acmpc_get_coefficients. */
#define ACMPC_COEF_STRIDE_SPATIAL 12  /* ds, a21, a31, b31, f3, v_ref, k_ref, ey_lo, ey_hi, 0, 0, 0 */
#define ACMPC_COEF_STRIDE_TEMPORAL 8  /* x, y, cos psi, sin psi, psi, k_ref, v_ref, w/2 - margin     */

/* Header floats of a winner record (see acmpc_record_floats). */
#define ACMPC_REC_COST 0
#define ACMPC_REC_VIOLATION 1
#define ACMPC_REC_NFEASIBLE 2
#define ACMPC_REC_OWNER 3
#define ACMPC_REC_HEADER 4

typedef struct acmpc_ctx acmpc_ctx;

/* Everything SpatialMPC.__init__ / ControlSolver.__setup / SpatialBicycleModel.__init__ hold as scalars
 * (spatial_mpc.py:21-58, control.py:108-158, dynamics.py:10-21). */
typedef struct acmpc_params {
  uint32_t struct_size;  /* sizeof(acmpc_params), for ABI checking                                   */
  int32_t mode;          /* ACMPC_MODE_*                                                              */
  int32_t device;        /* HIP device ordinal                                                        */
  int32_t max_problems;  /* capacity P (<= 65535)                                                     */
  int32_t max_candidates;/* capacity N (per problem)                                                  */
  int32_t max_steps;     /* capacity n (<= 1024)                                                      */
  int32_t nn_back;       /* mode T nearest-waypoint search: the W = nn_back + nn_ahead + 1 consecutive waypoints   */
  int32_t nn_ahead;      /*   from clamp(j_prev - nn_back, 0, n - W), j_prev = the previous step's nearest index (0
                              at the start); nn_ahead < 0 = exhaustive scan of all n waypoints (KDTree.query
                              semantics, localiser.py:282-289)                                                    */
  int32_t centre_update; /* acmpc_optimize: what the next round samples round - 0 = the round's argmin winner,
                            1 = the softmin-weighted mean of the round's candidates (weights exp(-(cost - min) /
                            softmin_lambda), the weighted-reduction form of localiser.py:572-579); in both cases the
                            best candidate so far stays in the pool, so a round never loses it                  */
  int32_t lq_candidate;  /* acmpc_optimize / acmpc_control_tick: 1 = the LAST sampling round's candidate 2 is the LQ plan -
                            the optimum of the reference's control QP (control.py:26-79) without its box rows, a backward
                            Riccati pass over the linearised model (dynamics.py:65-103), rolled forward with its feedback
                            and clipped into the input box (csrc/acmpc_lq.h; host, float64).  acmpc_optimize plans for the
                            paths of acmpc_set_paths and the x0 it is given; acmpc_control_tick, whose table is built on
                            the device, plans on the host meanwhile for this tick's path (acmpc_waypoint_table of `coords`)
                            and pose with this tick's speed profile as well (acmpc_velocity_ceiling +
                            acmpc_speed_profile_exact on that table: the prologue's own two passes; with qp_method 1, or an
                            infeasible profile, the one its PREVIOUS call solved).  With coords = NULL - the path cut
                            out of the map on the device - and a map index the host cuts the same window itself
                            (csrc/acmpc_prologue.h map_path_row: one statement for both sides); with a pose instead, whose
                            nearest map point the device searches, it plans for the previous call's problem as it was (none
                            in a handle's first tick then).  A first tick under qp_method 1 solves the speed profile
                            once on the host (acmpc_speed_profile_qp, cold) to plan with.  The argmin keeps the plan only
                            when it wins.
                            2 = as 1, and where that plan is not already the QP's optimum - a control on the input box, or
                            state rows (corridor control.py:57-60, t >= 0.01 control.py:134) violated beyond the solver's
                            acceptance tolerance - it is refined against the QP WITH its box rows (csrc/acmpc_lq_box.h: the
                            OSQP splitting with the model as a hard constraint of the Riccati z-update; host, float64,
                            iterate kept between calls; at most ACMPC_LQ_BOX_ITERATIONS = 40 iterations per call from a cold start, 12 from a kept iterate) and the
                            cheapest of {LQ plan, the two iterates} under J + w_bound V takes the slot.
                            0 = no such candidate */
  /* real-valued fields are doubles so that Python floats cross the ABI exactly; the device gets float32 */
  double step_cost[3];   /* Q  = diag(step_cost)  on (e_y, e_psi, t)     control.py:126               */
  double r_term[2];      /* R  = diag(r_term)     on (v, kappa)          control.py:127               */
  double final_cost[3];  /* QN = diag(final_cost)                        control.py:128               */
  double u_min[2];       /* QP input box: min_u - (0.1, 0)               control.py:130-139           */
  double u_max[2];       /*               max_u + (0.1, 0)                                           */
  double margin;         /* vehicle width / 2                            dynamics.py:14               */
  double wheelbase;      /* L                                            dynamics.py:11               */
  double t_min;          /* lower bound on the time state, 0.01          control.py:134               */
  double dt;             /* mode T Euler step [s]                                                     */
  double w_bound;        /* penalty weight on squared bound violation (build parameter, default 1e6)  */
  double softmin_lambda; /* temperature of the softmin-weighted mean (build parameter)                */
} acmpc_params;

/* Replaces: SpatialMPC.__init__ + ControlSolver.__setup (spatial_mpc.py:21-58, control.py:108-158).
 * Allocates host state only. */
int acmpc_create(const acmpc_params* params, acmpc_ctx** out);
void acmpc_destroy(acmpc_ctx* ctx);

/* The A/B switches of the tests and the tools (tools/README.md: ACMPC_NO_SOLO, ACMPC_TICK_GRAPH, ACMPC_SHAPE, ...) belong to
 * the handle: acmpc_create reads them from the environment ONCE, and this call sets one afterwards (`name` as the
 * environment spells it; value NULL, "" or - for the on / off ones - "0" restores the default).  No launch path reads the
 * environment.  ACMPC_EINVAL for an unknown name. */
int acmpc_set_option(acmpc_ctx* ctx, const char* name, const char* value);
/* One of them is for deployment rather than for A/B runs: "ACMPC_CONFORMANT_SYNC" = "1".  The default forms of the latency
 * paths lean on what gfx950 does rather than on what HIP and the HSA memory model promise (csrc/acmpc_kernels.hip, top):
 * values cross workgroups inside a launch as relaxed agent-scope atomics ordered by s_waitcnt vmcnt(0), waves of a
 * workgroup end while the others still meet at s_barrier, and acmpc_control_tick learns of completion from a flag the
 * last kernel writes into page-locked memory.  With the option on, every solve / round / batch is separate launches
 * ordered by the stream (rollout, then finalize), rounds run on one wave per workgroup, and completion is
 * hipStreamSynchronize - the same bits (tests/test_gpu_conformant.py), a few microseconds more per call
 * (INTEGRATION.md section 6).  Equivalent to ACMPC_NO_SOLO + ACMPC_NO_FUSED_FINALIZE + ACMPC_NO_CHAINED_STREAM +
 * ACMPC_TICK_NO_FLAG (and no ACMPC_TAILED_ROLLOUT). */

/* The LQ plan of acmpc_params::lq_candidate for one path, on the host (csrc/acmpc_lq.h): table [7][n] float64 in the
 * reference's row order, x0 = (e_y, e_psi, t), weights as in acmpc_params, the input box as the kernels hold it;
 * plan [n][2] float32 (v, kappa).  ACMPC_ESTATE when the problem has no finite plan.  No handle, no GPU work. */
int acmpc_lq_plan(const double* table, int32_t n, const double x0[3], const double step_cost[3], const double r_term[2],
                  const double final_cost[3], const float u_min[2], const float u_max[2], float* plan);

/* The plan of acmpc_params::lq_candidate = 2 for one path, on the host (csrc/acmpc_lq_box.h): arguments as acmpc_lq_plan,
 * `margin` / `w_bound` as in acmpc_params, at most `iterations` splitting iterations (12 when that is more and the iterate
 * handed in is a warm one).  `state` [1 + 8 n]: the iterate -
 * state[0] = n marks a warm one (wx, wu, lx, lu [n][2] each behind it), anything else starts cold; written back (state[0]
 * = 0 when the call kept none).  `info` [5]: iterations run (negative: a non-finite iterate), which plan took the slot
 * (0 the LQ plan, 1 the box iterate, 2 the clipped dynamics iterate), whether the refinement was triggered, and the
 * chosen plan's tracking cost J and summed squared state-row excess V (float64 rollout).  ACMPC_ESTATE when the
 * problem has no finite LQ plan.  No handle, no GPU work. */
int acmpc_lq_box_plan(const double* table, int32_t n, const double x0[3], const double step_cost[3], const double r_term[2],
                      const double final_cost[3], const float u_min[2], const float u_max[2], double margin, double w_bound,
                      int32_t iterations, double* state, float* plan, double* info);
/* What the handle's last lq_candidate = 2 plan did: `info` [5] as acmpc_lq_box_plan's. */
int acmpc_lq_box_stats(const acmpc_ctx* ctx, double info[5]);

/* Message of the last failing call on `ctx` (or of the last failing acmpc_create when ctx is NULL). */
const char* acmpc_last_error(const acmpc_ctx* ctx);

/* Replaces: ControlSolver._update_references + _update_problem_bounds (control.py:26-33,47-70), i.e.
 * SpatialBicycleModel.linearise (dynamics.py:65-103) and the corridor bounds, for P reference paths at once.
 * `tables` is P consecutive 7 x n float64 ReferencePath arrays in the reference's row order
 * [x, y, psi, kappa, ds, width, v] (control/paths.py:4-72).  Coefficients are computed in float64 on the host,
 * stored as float32, and uploaded by the next device call.  No device work. */
int acmpc_set_paths(acmpc_ctx* ctx, const double* tables, int32_t P, int32_t n);

/* The packed float32 tables themselves ([P][n][ACMPC_COEF_STRIDE_* by the handle's mode], the layout
 * acmpc_get_coefficients and acmpc_tick_read_device_tables hand out) instead of paths: for callers that keep their
 * tables packed, and for the tests, which feed a tick's own device-built table to the two-call path so that
 * `acmpc_control_tick == acmpc_set_paths + acmpc_optimize` holds bit for bit whatever the last float32 bit of a host
 * cos / sin was.  CONTRACT for handles with lq_candidate != 0: the LQ plan (candidate 2 of the last round) is computed from
 * float64 paths, which this call does not carry - the float64 tables of an earlier acmpc_set_paths of the SAME shape (P, n)
 * stay and are taken to describe the same paths as `coef` (the caller's responsibility: the packed table is meant to be
 * that acmpc_set_paths' own, re-packed); with no such tables, or after a call with another shape, acmpc_optimize runs
 * without the plan (candidate 2 is then an ordinary sample).  Call acmpc_set_paths again to plan for new paths. */
int acmpc_set_coefficients(acmpc_ctx* ctx, const float* coef, int32_t P, int32_t n);

/* Copies the packed float32 table of problem `problem` (n rows of ACMPC_COEF_STRIDE_* floats) to `out`.
 * Host only; lets CPU tests pin the host-side arithmetic against the oracle. */
int acmpc_get_coefficients(const acmpc_ctx* ctx, int32_t problem, float* out, int32_t capacity_floats);

/* Floats in one winner record: ACMPC_REC_HEADER + 2 n + 3 (n + 1) =
 *   [cost, violation, n_feasible, owner, u_0 .. u_{n-1} (2 each), x_0 .. x_n (3 each)]
 * x is (e_y, e_psi, t) in mode S and (X, Y, phi) in mode T.  The u / x blocks are what
 * spatial_mpc.py:193-202 slices out of OSQP's dec.x ([x_0..x_n ; u_0..u_{n-1}], control.py:121-158). */
int32_t acmpc_record_floats(int32_t n);

/* Replaces: ControlSolver.solve (control.py:15-24) for host-resident inputs - one blocking call doing H2D,
 * rollout + cost + argmin, and D2H.
 *   x0        [P][3]   mode S: t2s(...) output (e_y, e_psi, t) (dynamics.py:23-40); mode T: pose (X, Y, phi)
 *   U         control-sample matrix in `layout`
 *   costs     [P][N] or NULL
 *   best_idx  [P] index of the cheapest candidate (lowest index on ties; non-finite costs rank last)
 *   records   [P][acmpc_record_floats(n)] or NULL
 */
int acmpc_solve(acmpc_ctx* ctx, const float* x0, const float* U, int32_t P, int32_t N, int32_t n,
                int32_t layout, float* costs, int32_t* best_idx, float* records);

/* Page-locked host memory for acmpc_solve's control matrix.  A matrix in ordinary (pageable) memory is staged into
 * device memory by a copy in front of the rollout (~40 us for the 1.6 MB of 4 096 candidates x horizon 50); one built in
 * memory these calls return is READ IN PLACE by the rollout, which streams it over the host link while it computes - a
 * 4 096-candidate solve 58 us instead of 74 (measured through the Python wrapper, tools/archive/time_host_solve.py).  Start states,
 * keys and records of a one-launch solve never travel as copies of their own either way: the kernels read and write them
 * in the handle's page-locked block.  acmpc_host_alloc initialises the HIP runtime: call it in the process that solves
 * (after the fork of controller.py:94-100), never before.  The closed loop does not need it: acmpc_control_tick samples
 * on the device and moves 2 kB per tick. */
int acmpc_host_alloc(void** out, uint64_t bytes);
int acmpc_host_free(void* memory);

/* Same work with every buffer already resident in device memory, asynchronous on `stream`
 * (a hipStream_t, NULL = the null stream).  `d_keys` [P] receives the packed (cost, index) keys - see
 * acmpc_rollout_device - and `d_records` [P][acmpc_record_floats(n)] the winner records.
 * Device buffers are what hipMalloc returns, or views into it on their element type's boundary (gfx950 serves a
 * 16-byte load on any 4-byte boundary); buffers that start 16-byte aligned are served fastest - e.g. a candidate-major
 * `d_U` then takes the faster of the two tile kernels, any other start the slower one, with the same results. */
int acmpc_solve_device(acmpc_ctx* ctx, const float* d_x0, const float* d_U, int32_t P, int32_t N, int32_t n,
                       int32_t layout, float* d_costs, int64_t* d_keys, float* d_records, void* stream);

/* Sharded form: each rank rolls out its own N candidates whose global indices start at
 * `index_offset`, and gets per-problem keys
 *     key = (ordered_int32(cost) << 32) | uint32(global index)
 * that order like (cost, index) under signed 64-bit comparison, so ONE RCCL all-reduce(MIN) over `d_keys`
 * yields the global argmin with lowest-index tie-breaking.  acmpc_finalize_device then writes the winner's
 * record on the rank that owns it and zeros elsewhere (owner flag 0), plus every rank's feasible count, so an
 * all-reduce(SUM) of the records gives every rank the selected controls.
 * `d_keys` may be NULL in both calls on a single GPU: the rollout then leaves its per-workgroup partial keys in
 * the handle and the finalize step reduces those itself (two launches in total; acmpc_solve_device does the same
 * work in ONE launch for mode S problems of up to 65 536 candidates in all, and in these two launches otherwise). */
int acmpc_rollout_device(acmpc_ctx* ctx, const float* d_x0, const float* d_U, int32_t P, int32_t N, int32_t n,
                         int32_t layout, int64_t index_offset, float* d_costs, int64_t* d_keys, void* stream);
int acmpc_finalize_device(acmpc_ctx* ctx, const int64_t* d_keys, const float* d_x0, const float* d_U, int32_t P,
                          int32_t N, int32_t n, int32_t layout, int64_t index_offset, float* d_records,
                          void* stream);

/* Softmin-weighted mean control sequence, sum_c w_c U_c / sum_c w_c with w_c = exp(-(cost_c - min)/lambda)
 * (the weighted-reduction form of localiser.py:572-579).  `d_costs` [P][N] and `d_keys` [P] come from a
 * previous rollout; `d_mean` [P][n][2] (float32), `d_weight_sum` [P] (float64) or NULL.  Partial sums are
 * combined in a fixed order: results are bitwise reproducible. */
int acmpc_softmin_device(acmpc_ctx* ctx, const float* d_costs, const int64_t* d_keys, const float* d_U,
                         int32_t P, int32_t N, int32_t n, int32_t layout, float* d_mean, double* d_weight_sum,
                         void* stream);

/* Uploads pending tables now (otherwise done by the next device call) and blocks until resident, so that a
 * timed region contains no host-to-device traffic. */
int acmpc_sync_tables(acmpc_ctx* ctx, void* stream);

/* On-device candidate generation.  Candidate c of problem p is
 *     U_c = clip(centre_p + a_c * (sigma_v, sigma_kappa) * smooth_noise_c, input box),   a_c = ((c mod 8) + 1) / 8,
 * smooth_noise = raised-cosine blend along the horizon of 8 x 2 standard normals drawn with Philox4x32-10 at
 * counter (global candidate index, problem, round, draw) and key = seed, so every rank regenerates the same
 * candidate from its index alone.  Candidate 0 is the centre itself, candidate 1 is `d_u_ref` when given.
 * `d_centre` holds P rows of `centre_stride` floats whose first 2n are (v, kappa) per step - the u block of a
 * winner record qualifies (centre_stride = acmpc_record_floats(n), pointer = records + ACMPC_REC_HEADER). */
int acmpc_sample_device(acmpc_ctx* ctx, const float* d_centre, int32_t centre_stride, const float* d_u_ref,
                        int32_t P, int32_t N, int32_t n, int32_t layout, int64_t index_offset, double sigma_v,
                        double sigma_kappa, uint64_t seed, uint32_t round, float* d_U, void* stream);

/* Finalize for sampled candidates: the winner's controls are re-drawn from the global index in its key (same
 * arithmetic as acmpc_sample_device, bit-identical), so after ONE all-reduce(MIN) of `d_keys` every rank writes
 * the complete winner record itself - no second collective, no owner.  `d_keys` may be NULL on a single GPU
 * (the handle's partial keys of the preceding rollout are reduced instead).  n_feasible in the record is this
 * rank's count.  Arguments after `d_x0` must be those the candidates were sampled with. */
int acmpc_finalize_sampled_device(acmpc_ctx* ctx, const int64_t* d_keys, const float* d_x0, const float* d_centre,
                                  int32_t centre_stride, const float* d_u_ref, int32_t P, int32_t N, int32_t n,
                                  double sigma_v, double sigma_kappa, uint64_t seed, uint32_t round,
                                  float* d_records, void* stream);

/* acmpc_rollout_device + acmpc_finalize_sampled_device as ONE call for a rank that has all the candidates (no all-reduce
 * between them): `d_U` holds what acmpc_sample_device wrote for (centre, u_ref, sigma, seed, round); rollout, argmin and
 * the winners' records - re-drawn from their indices - without a host round trip between them: two launches, or with
 * the handle's ACMPC_TAILED_ROLLOUT option one (mode S, step-major: the last workgroup of every problem finalizes it,
 * csrc/acmpc_kernels.hip rollout_tailed_kernel - measured slower on the headline's batch, so not the default).
 * d_costs [P][N] or NULL, d_keys [P] or NULL, d_records [P][acmpc_record_floats(n)].  The same bits either way. */
int acmpc_solve_sampled_device(acmpc_ctx* ctx, const float* d_x0, const float* d_U, const float* d_centre,
                               int32_t centre_stride, const float* d_u_ref, int32_t P, int32_t N, int32_t n, int32_t layout,
                               double sigma_v, double sigma_kappa, uint64_t seed, uint32_t round, float* d_costs,
                               int64_t* d_keys, float* d_records, void* stream);

/* A STREAM of batches on one rank: acmpc_solve_device (d_centre NULL: the winners are read from d_U) or
 * acmpc_solve_sampled_device (d_centre given: re-drawn) with the argmin and the winners' records of one call DEFERRED -
 * they are computed inside the NEXT call's rollout launch (csrc/acmpc_kernels.hip rollout_chained_kernel: one row in
 * every few of the grid finalizes four problems of the previous batch per workgroup while the other rows stream), or by
 * acmpc_solve_stream_flush behind the last one.  Behind the headline's rollout the finalize of 4 096 problems is a launch
 * of its own - 13 us and a launch boundary per 1.0 ms step; inside the next rollout nobody waits for it.
 *   d_keys / d_records of call k are complete once call k + 1 on the same stream, or the flush, has completed;
 *   until then the caller leaves call k's d_x0, d_centre / d_u_ref - and d_U when d_centre is NULL - as they were;
 *   d_costs of call k are complete with call k itself.
 * The same bits as acmpc_solve_device / acmpc_solve_sampled_device (tests/test_gpu_tailed_rollout.py).  Shapes the one
 * launch does not take (mode T, the candidate-major layout, horizons whose finalize would cost the rollout its
 * occupancy) run the pending finalize as a launch of its own in front of the rollout.  While a batch is pending every
 * other solve on the handle returns ACMPC_ESTATE (acmpc_sample_device - drawing the next batch's candidates - is allowed);
 * new tables (acmpc_set_paths) are uploaded behind the pending finalize. */
int acmpc_solve_stream_device(acmpc_ctx* ctx, const float* d_x0, const float* d_U, const float* d_centre,
                              int32_t centre_stride, const float* d_u_ref, int32_t P, int32_t N, int32_t n, int32_t layout,
                              double sigma_v, double sigma_kappa, uint64_t seed, uint32_t round, float* d_costs,
                              int64_t* d_keys, float* d_records, void* stream);
int acmpc_solve_stream_flush(acmpc_ctx* ctx, void* stream);

/* The one collective of the multi-GPU step, for hosts that drive RCCL themselves rather than through
 * torch.distributed: in-place all-reduce(MIN) of the P packed keys over `rccl_comm` (an `ncclComm_t` the host
 * created, one rank per GPU), enqueued on `stream`.  Call between acmpc_rollout_device and
 * acmpc_finalize_sampled_device / acmpc_finalize_device.  RCCL is resolved at the first call: the copy already loaded
 * in the process if there is one (the one that owns `rccl_comm`), else `librccl.so.1` (or $ACMPC_RCCL_LIBRARY);
 * the library itself does not link RCCL.  ACMPC_ESTATE when no RCCL can be found, ACMPC_EHIP when RCCL fails. */
int acmpc_reduce_across_ranks(acmpc_ctx* ctx, void* rccl_comm, int64_t* d_keys, int32_t P, void* stream);
/* A communicator for it out of the same copy of RCCL (a process may hold two - the system's and the one PyTorch bundles -
 * and a communicator only works with the copy that made it): acmpc_rccl_unique_id on ONE rank (ncclGetUniqueId; 128
 * bytes, handed to the others by whatever channel the host has - bench.py uses the torch.distributed store),
 * acmpc_rccl_comm_create on EVERY rank (ncclCommInitRank: collective, returns when all `n_ranks` have called it; `device`
 * >= 0 is made current first, one rank per GPU), acmpc_rccl_comm_destroy at the end.  ACMPC_ESTATE when no RCCL can be
 * found, ACMPC_EHIP when RCCL fails, ACMPC_ENODEVICE for a device that cannot be selected.  No handle. */
#define ACMPC_RCCL_UNIQUE_ID_BYTES 128
int acmpc_rccl_unique_id(void* id_out);
int acmpc_rccl_comm_create(const void* id, int32_t n_ranks, int32_t rank, int32_t device, void** comm_out);
int acmpc_rccl_comm_destroy(void* comm);

/* Replaces: ControlSolver.solve (control.py:15-24) end to end on the device - `rounds` rounds of
 * sample -> rollout + cost -> argmin, each round sampling round the previous winner with the spread shrunk by
 * `shrink`, one host round trip in total (x0, centre, u_ref up; the final records down).  Host pointers:
 * x0 [P][3], centre [P][n][2], u_ref [P][n][2] or NULL, sigma[2], records [P][acmpc_record_floats(n)]. */
int acmpc_optimize(acmpc_ctx* ctx, const float* x0, const float* centre, const float* u_ref, int32_t P, int32_t N,
                   int32_t n, int32_t rounds, const double sigma[2], double shrink, uint64_t seed, float* records);

/* Replaces: the whole of SpatialMPC.get_control (spatial_mpc.py:170-217) for one tick of the control loop, as ONE host
 * round trip: the H x 3 reference path, the constraints and the centre sequence go up in one copy; the device runs
 *   prologue (one wavefront): construct_waypoints (spatial_mpc.py:125-154) -> velocity ceiling + speed-profile QP
 *     (speed_profile.py:26-59, tridiagonal ADMM warm-started from the state the handle keeps on the device) -> t2s
 *     (dynamics.py:23-40) -> linearise + corridor rows (dynamics.py:65-103, control.py:57-60) -> reference controls,
 *   `rounds` rounds of sample -> rollout + cost -> argmin (as acmpc_optimize),
 * all nodes of one captured hipGraph; the winner's record, the 7 x n table and the QP status come back through pinned
 * host memory, and the tail of get_control (acmpc_unpack_decision) runs before the call returns.  Handles with
 * centre_update = 0 and horizon - 1 <= 128 only (ACMPC_ESTATE otherwise: use acmpc_set_paths + acmpc_optimize).
 * Mode T handles (the Cartesian rollout with nearest-waypoint projection, north_star's literal shape) take the same
 * call: the prologue then leaves the pose (offset, 0, pi / 2) as the start state and the [n][8] waypoint rows as the
 * table, the rounds roll them with the handle's search window (nn_back / nn_ahead; exhaustive when nn_ahead < 0) and
 * the plan is unpacked by acmpc_unpack_decision_temporal. */
typedef struct acmpc_tick {
  uint32_t struct_size;        /* sizeof(acmpc_tick)                                                            */
  int32_t horizon;             /* H = rows of `coords`; n = H - 1                                               */
  int32_t localised;           /* != 0: LocalisedSpeedProfileSolver (ceiling = v_max everywhere, no end velocity) */
  int32_t has_end_velocity;    /* != 0: last ceiling entry = end_velocity (speed_profile.py:42-43)              */
  int32_t n_candidates;        /* N per round                                                                   */
  int32_t rounds;
  int32_t centre_is_reference; /* != 0: sample round the reference controls (`centre` is ignored, may be NULL)  */
  int32_t qp_max_iter;         /* speed-profile QP: iteration cap (the reference passes 4000, spatial_mpc.py:17) */
  int32_t qp_check_every;      /* stopping test every this many iterations (<= 0: 10)                           */
  int32_t qp_method;           /* speed-profile QP: 0 = its exact optimum in two passes (acmpc_speed_profile_exact), the
                                  splitting below only where that does not apply (an infeasible or misshapen problem);
                                  1 = always the OSQP splitting (acmpc_speed_profile_qp), warm-started between ticks    */
  double offset;               /* lateral displacement of the car: pose (offset, 0, pi/2), spatial_mpc.py:187   */
  double v_min, v_max, a_min, a_max, ay_max, ki_min, end_velocity; /* speed_profile_constraints, read every tick */
  double sigma[2];             /* first round's spread of (v, kappa); round r uses sigma * shrink^r             */
  double shrink;
  double qp_eps_abs, qp_eps_rel;
  uint64_t seed;
  /* reference path from the map bound with acmpc_bind_map instead of from `coords` (pass coords = NULL) */
  int32_t map_index;           /* first map waypoint of the look-ahead window; < 0: the map point nearest to the pose  */
  int32_t centreline_points;   /* length of the resampled centre line (500, controller.py:102-108); multiple of horizon */
  double pose_x, pose_y;       /* map frame; read when map_index < 0                                                  */
  double lateral_offset;       /* subtracted from the window's lateral coordinate (the car's offset from the line)    */
} acmpc_tick;

/* Replaces: the path from the map to get_control's argument for a pose on a known map - nearest map
 * point (what the localiser's KD-tree query returns, localiser.py:282-289), the next 150 m of centre line
 * (perception/tracks.py:14) in the vehicle frame, resampled to `centreline_points` float32 points (controller.py:102-108),
 * every (points / H)-th kept with widths linspace(10, 6, H) (ControlProcess._reference_path, controller.py:256-267).
 * acmpc_bind_map copies the centre polyline [M][2] float64 (utils/load.py:9-35 order x, y; `spacing` = metres between map
 * points, mapping/map_maker.py:203); no device work.  acmpc_control_tick with coords = NULL then runs the window kernel
 * in front of the prologue; acmpc_map_reference_path runs it alone (blocking; coords [H][3] out, *first_index out). */
int acmpc_bind_map(acmpc_ctx* ctx, const double* centre, int32_t M, double spacing);
int acmpc_map_reference_path(acmpc_ctx* ctx, int32_t map_index, double pose_x, double pose_y, double lateral_offset,
                             int32_t horizon, int32_t centreline_points, double* coords, int32_t* first_index);

/* coords [H][3] float64 (x, y, width) in the vehicle frame, or NULL with a bound map (then `coords_out` [H][3], if not
 * NULL, receives the path the device built); centre [n][2] float32 (v, kappa) or NULL.
 * Outputs (all required): table [7][n] float64 (rows x, y, psi, kappa, ds, width, v); record
 * [acmpc_record_floats(n)]; decision [5n + 3] = dec.x ([x_0..x_n ; u_0..u_{n-1}], control.py:121-158);
 * projected_control [2][n], prediction [n][2], cum_time [n], times / accelerations / steer_rates [n - 1] as
 * acmpc_unpack_decision writes them; info [8] = {cost, violation, n_feasible, max |dec.x|, QP status (0 = solved,
 * 1 = maximum iterations), QP iterations, first map index of the window or -1, 1 when the cost, the violation or an
 * entry of the plan is not finite (the reference's solver would not report such a solve as solved) else 0}. */
int acmpc_control_tick(acmpc_ctx* ctx, const acmpc_tick* tick, const double* coords, const float* centre, double* table,
                       float* record, double* decision, double* projected_control, double* prediction,
                       double* cum_time, double* times, double* accelerations, double* steer_rates, double* info,
                       double* coords_out);

/* Test hooks of the tick path.  acmpc_tick_read_device_tables copies what the last tick's prologue left on the device
 * for the rollout - x0 [3], u_ref [n][2], the packed table [n][ACMPC_COEF_STRIDE_SPATIAL or _TEMPORAL, by the handle's
 * mode] - back to the host.
 * acmpc_tick_read_device_frames: the frames of the verified nearest-waypoint search the prologue tabulated for that
 * table (mode T handles with the exhaustive search; acmpc_search_frame_floats(n) floats) - the same arithmetic as
 * acmpc_search_frames, run by the prologue's lanes.
 * acmpc_speed_profile_qp_device runs the prologue's ADMM alone on the GPU (host pointers, blocking): same arguments
 * and, bit for bit, the same results as acmpc_speed_profile_qp. */
int acmpc_tick_read_device_tables(acmpc_ctx* ctx, float* x0, float* u_ref, float* coef);
int acmpc_tick_read_device_frames(acmpc_ctx* ctx, float* out, int64_t capacity_floats);
int acmpc_speed_profile_qp_device(acmpc_ctx* ctx, const double* v_hi, const double* ds, int32_t n, double a_min,
                                  double a_max, double v_min, int32_t max_iter, int32_t check_every, double eps_abs,
                                  double eps_rel, double* v, double* y, int32_t warm_start, int32_t* iterations);

/* Host-side float64 helpers round one solve (csrc/acmpc_host_path.cpp; no GPU work, no handle).
 * Replaces: SpatialMPC.construct_waypoints (spatial_mpc.py:125-154).  coords [H][3] = (x, y, width) ->
 * table [7][n], n = H - 1, rows [x, y, psi, kappa, ds, width, v = 0]. */
int acmpc_waypoint_table(const double* coords, int32_t H, double eps, double* table);
/* Replaces: the v_max vector of SpeedProfileSolver / LocalisedSpeedProfileSolver (speed_profile.py:26-43,131-150):
 * max(v_min, min(sqrt(ay_max / (|kappa| + 1e-12)), v_max)) + 2 with v_max where |kappa| < ki_min, last entry =
 * end_velocity when given; `localised` != 0: v_max everywhere. */
int acmpc_velocity_ceiling(const double* kappa, int32_t n, double ay_max, double ki_min, double v_min, double v_max,
                           int32_t localised, int32_t has_end_velocity, double end_velocity, double* ceiling);
/* Replaces: the tail of SpatialMPC.get_control (spatial_mpc.py:156-168,195-211).  z = dec.x ([x_0..x_n ; u_0..u_{n-1}],
 * 5n + 3 doubles), table [7][n] -> projected_control [2][n] = (v, atan(kappa L)), prediction [n][2] (s2t),
 * cum_time [n], times / accelerations / steer_rates [n - 1]. */
int acmpc_unpack_decision(const double* z, int32_t n, const double* table, double wheelbase, double* projected_control,
                          double* prediction, double* cum_time, double* times, double* accelerations,
                          double* steer_rates);
/* The same for a mode T plan (build-defined: the reference has no temporal rollout).  z's states are the poses
 * (X, Y, phi) after each Euler step of `dt` seconds: prediction [n][2] = the first n of them, cum_time [i] = i dt,
 * times = dt, and the derivatives are those of the plan's own controls: accelerations = dv / dt, steer_rates =
 * d(delta) / dt. */
int acmpc_unpack_decision_temporal(const double* z, int32_t n, double dt, double wheelbase, double* projected_control,
                                   double* prediction, double* cum_time, double* times, double* accelerations,
                                   double* steer_rates);

/* Host side of mode T's verified nearest-waypoint search (exhaustive semantics, localiser.py:282-289: the first minimum
 * over ALL waypoints), as acmpc_set_paths tabulates it for the kernels: coef = P packed [n][8] waypoint rows
 * (acmpc_get_coefficients) -> per problem acmpc_search_frame_floats(n) floats: for every window of
 * acmpc_search_window(NULL) consecutive waypoints [t_x, t_y, k_along, k_across, slab, tube, far, -slack].  The kernel
 * accepts the window's first minimum j as the global one when
 *     |fma(Y, Y, fma(X, X, key_j))| < min(fma(across, across, fma(along, along, -slack)), far),
 * alpha = fma(t_x, X, fma(t_y, Y, k_along)), beta = fma(t_x, Y, fma(-t_y, X, k_across)), along = med3(alpha,
 * slab - alpha, 0), across = med3(|beta| - tube, 0, 32), and scans the whole path otherwise.  No device work: lets
 * tests check on the CPU that an accepted index is always the exhaustive one.  n < the window: ACMPC_EINVAL. */
int32_t acmpc_search_window(int32_t* back);
int32_t acmpc_search_frame_floats(int32_t n);
int acmpc_search_frames(const float* coef, int32_t P, int32_t n, float* out, int64_t capacity_floats);

/* The generator itself, on the host (same code as the kernels): lets tests pin the integer stream. */
void acmpc_philox4x32(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);

/* Replaces: the osqp calls of SpeedProfileSolver (src/acmpc/control/solvers/speed_profile.py:61-86).  Solves
 *     min 1/2 |v|^2 - v_hi'v   s.t.  a_min <= (v[i+1] - v[i]) / (2 ds[i]) <= a_max,  v_min <= v <= v_hi
 * on the host with the OSQP splitting specialised to the problem's tridiagonal structure: O(n) per iteration, so the
 * whole-lap profile (n ~ 1e4, spatial_mpc.py:60-87) is as cheap per iteration as the horizon's.  `v` [n] and `y`
 * [2n - 1] hold the primal/dual iterate: read when warm_start != 0, always written.  The stopping test (OSQP's, at
 * eps_abs / eps_rel) runs every `check_every` iterations (<= 0: 10).  Returns 0 = solved, 1 = maximum iterations
 * reached, ACMPC_EINVAL on bad arguments.  No GPU work.  The device prologue of acmpc_control_tick runs the same
 * statement of the algorithm (csrc/acmpc_admm.h) on one wavefront: same float64 operations in the same order. */
int acmpc_speed_profile_qp(const double* v_hi, const double* ds, int32_t n, double a_min, double a_max, double v_min,
                           int32_t max_iter, int32_t check_every, double eps_abs, double eps_rel, double* v, double* y,
                           int32_t warm_start, int32_t* iterations);

/* The same QP solved EXACTLY, without iterating (csrc/acmpc_admm.h exact_profile): its objective is 1/2 |v - v_hi|^2 up to a
 * constant and v_hi is also the upper bound, so the optimum is the pointwise largest feasible profile - v_hi cut down by a
 * forward sweep (a_max) and a backward sweep (a_min).  Returns 0 with v [n] = the optimum and y [2n - 1] = 0; 1 when the
 * problem is not of that shape (a_min > 0, a_max < 0, a spacing that is not positive and finite, a non-finite ceiling) or
 * infeasible (some v below v_min): solve it with acmpc_speed_profile_qp then, whose status is the reference's for such a
 * problem; ACMPC_EINVAL on bad arguments.  acmpc_control_tick's prologue does exactly this on the device (qp_method 0).
 * No GPU work. */
int acmpc_speed_profile_exact(const double* v_hi, const double* ds, int32_t n, double a_min, double a_max, double v_min,
                              double* v, double* y);

/* Diagnostic: with acmpc_set_option(ctx, "ACMPC_START_CLOCKS", "1") every workgroup of a rollout launch
 * (acmpc_rollout_device and the calls built on it; not the candidate-major tile kernels) leaves the 100 MHz wall clock of
 * its first instruction on the device; this copies the LAST launch's out (after a device synchronise): `count` = its
 * workgroups, P x blocks per problem, problem-major.  A launch that fits the chip in one generation should start all of
 * them within a couple of microseconds; workgroups that start tens of microseconds late are what a dispatcher that
 * over-subscribes some compute units looks like.  *count = 0: no stamped launch yet. */
int acmpc_rollout_start_clocks(acmpc_ctx* ctx, uint64_t* out, int32_t capacity, int32_t* count);

/* Measurement hooks.  After acmpc_profile_enable(ctx, K) the next K rollout launches of this handle carry a HIP
 * event pair attached to the dispatch itself (hipExtLaunchKernel: the kernel's own begin/end timestamps on the
 * stream it is launched on, no marker packets between launches); acmpc_profile_collect waits for them, writes the
 * per-launch durations in milliseconds and re-arms the K pairs.  K = 0 disables. */
int acmpc_profile_enable(acmpc_ctx* ctx, int32_t capacity);
int acmpc_profile_collect(acmpc_ctx* ctx, float* out_ms, int32_t capacity, int32_t* count);

/* Key helpers (host side; same packing as the kernels). */
int64_t acmpc_pack_key(float cost, uint32_t index);
float acmpc_key_cost(int64_t key);
uint32_t acmpc_key_index(int64_t key);

/* ---- particle-filter localiser: scoring on the GPU -------------------------------------
 * Replaces the data-parallel part of LocalisationProcess._update_particles (src/acmpc/localisation/localiser.py:
 * 255-410): three nearest-point queries per particle against the centre / left / right map polylines (what the
 * reference's scipy KD-trees answer), the heading offset, the observed track limits placed in every particle's
 * frame against the map limits ahead of it, the Gaussian score and the validity mask (localiser.py:453-462).
 * Resampling (sequential, random) stays with the caller.  Same conventions as above; no device work at create. */
typedef struct acmpc_pf acmpc_pf;

typedef struct acmpc_pf_params {
  uint32_t struct_size;
  int32_t device;                   /* HIP device ordinal, -1 = current                                      */
  int32_t max_particles;            /* localisation.n_particles (configs/monza.yaml:47)                       */
  int32_t max_observation_points;   /* left + right observed limit points after downsampling                 */
  double score_mean, score_sigma;   /* localisation.score_distribution (monza.yaml:61-63)                     */
  double threshold_rotation;        /* [rad]  localisation.thresholds.rotation * pi / 180 (localiser.py:616)  */
  double threshold_offset;          /* [m]    localisation.thresholds.offset                                  */
  double threshold_error;           /* [m]    localisation.thresholds.track_limit                             */
  double wheelbase;                 /* for acmpc_pf_advance (localiser.py:83, 94)                             */
} acmpc_pf_params;

/* centre / left / right: map polylines [m][2] float64 (utils/load.py:9-35 order x, y); copied. */
int acmpc_pf_create(const acmpc_pf_params* params, const double* centre, int32_t m_centre, const double* left,
                    int32_t m_left, const double* right, int32_t m_right, acmpc_pf** out);
void acmpc_pf_destroy(acmpc_pf* handle);
const char* acmpc_pf_last_error(const acmpc_pf* handle);
double acmpc_pf_score_scale(const acmpc_pf* handle); /* max pdf over linspace(-10, 10, 100), localiser.py:655-661 */

/* states [P][3] float32 (x, y, yaw); obs_left [k_left][2], obs_right [k_right][2] float32 in the vehicle frame
 * (x right, y forward), already downsampled and cut at y < 50 (localiser.py:246-253,336-337).  Outputs per particle:
 * track_indices [P][3] (centre, left, right), minimum_offset, heading_offset, observation_error, score (float64 as
 * in the reference), valid (0/1).  Blocking; host pointers. */
int acmpc_pf_score(acmpc_pf* handle, const float* states, int32_t P, const float* obs_left, int32_t k_left,
                   const float* obs_right, int32_t k_right, int32_t* track_indices, double* minimum_offset,
                   double* heading_offset, double* observation_error, double* score, uint8_t* valid);

/* states += (v cos phi, v sin phi, v tan(delta) / L) * dt with per-particle delta, v (localiser.py:66-95). */
int acmpc_pf_advance(acmpc_pf* handle, float* states, const float* delta, const float* velocity, int32_t P, double dt);

/* Score-weighted mean state with the NaN -> uniform fallback, and the largest distance / yaw difference of any
 * particle to it - the two numbers the convergence flag compares with its limits (localiser.py:561-579). */
int acmpc_pf_estimate(acmpc_pf* handle, const float* states, const float* scores, int32_t P, double estimate[3],
                      double* max_distance, double* max_angle);

/* ---- device-resident particle filter: one host round trip per update ------------------------------------------------
 * The reference's update cycle with the particles living on the GPU: Localiser.step (localiser.py:41-77: every particle
 * moves with its own noisy control), then _score_particles (localiser.py:234-239): score, publish the scores, resample
 * (keep the valid particles in order; fewer than `minimum_particles` left -> _reset_filter, localiser.py:468-485;
 * otherwise top up to `n_desired` with copies of kept particles drawn in proportion to their score, plus Gaussian
 * noise, localiser.py:486-545), estimate and convergence numbers (localiser.py:561-579).  The random draws are
 * Philox4x32-10 at counter (index, `counter`, tag, draw) with key `seed` - NOT NumPy's global stream, whose call order a
 * kernel cannot follow: the host-side ParticleFilter (NumPy resampling, the reference's draw order) stays the parity
 * mode, this path is pinned by the oracle's restatement of these draws (exact picked indices: integer weights
 * floor(score 2^40), integer prefix sums, mulhi of a 64-bit word with the total). */
typedef struct acmpc_pf_resample {
  uint32_t struct_size;
  int32_t n_desired;          /* n_particles, or n_converged_particles once converged (localiser.py:497-500)         */
  int32_t minimum_particles;  /* thresholds.minimum_particles                                                       */
  uint32_t counter;           /* update number: a fresh set of draws per update                                     */
  uint64_t seed;
  double sigma_x, sigma_y, sigma_yaw;  /* sampling_noise (yaw in radians)                                           */
} acmpc_pf_resample;

/* _reset_filter on the device: `n` particles evenly along the centre line, uniform scores. */
int acmpc_pf_filter_reset(acmpc_pf* handle, int32_t n);
/* Upload / download the live particles (states [n][3], scores [n] float32). */
int acmpc_pf_filter_set(acmpc_pf* handle, const float* states, const float* scores, int32_t n);
int acmpc_pf_filter_get(acmpc_pf* handle, float* states, float* scores, int32_t capacity, int32_t* n);
/* Localiser.step: delta = tyre_angle + N(0, sigma_yaw), speed = |velocity + N(0, sigma_velocity)|, kinematic Euler step.
 * Asynchronous (ordered before the next update on the handle's stream). */
int acmpc_pf_filter_step(acmpc_pf* handle, double tyre_angle, double velocity, double dt, double sigma_yaw,
                         double sigma_velocity, uint64_t seed, uint32_t counter);
/* One _score_particles.  result [8] = {estimate x, y, yaw, max distance, max |yaw difference| to the estimate,
 * live particles after the update, valid particles of this scoring, 1 if the filter was reset}. */
int acmpc_pf_filter_update(acmpc_pf* handle, const float* obs_left, int32_t k_left, const float* obs_right,
                           int32_t k_right, const acmpc_pf_resample* resample, double* result);

/* Library identification: "acmpc-hip <version> gfx950". */
const char* acmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ACMPC_H */
