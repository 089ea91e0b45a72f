// gfx950 kernels of the ac-mpc rollout-and-cost path.
//
//   rollout_kernel   one lane = one candidate (or CPT adjacent candidates), steps sequential; the per-step
//                    table is wave-uniform, so in mode S it is read with scalar loads (SGPRs, no LDS traffic)
//                    and in mode T - where every lane gathers "its" nearest waypoint - it is staged in LDS once
//                    per workgroup.  Controls are streamed from HBM exactly once; costs are written once.
//                    Each workgroup reduces its (cost, index) keys with wave shuffles + LDS and writes ONE
//                    partial key: no atomics, no pre-zeroed buffers, bitwise reproducible.
//   finalize_kernel  one wave per problem: min over the partial keys (or takes all-reduced keys), re-rolls the
//                    winning candidate and writes its record [cost, violation, n_feasible, owner, u, x].
//   softmin_*        score-weighted mean of the control sequences.
//
// Built with -ffp-contract=off: see acmpc_device.h.
#include "acmpc_kernels.h"

#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

// gfx950 (MI355X) only, on purpose.  Three things in this file lean on what that hardware does rather than on what HIP
// promises, and must not be compiled for anything else without being revisited:
//   - the multi-wave rounds and the one-launch solve let waves of a workgroup END while the others keep meeting at
//     s_barrier (rollout_sampled_trio / quad / pair kernels, rollout_solo_kernel<SPLIT>): the hardware takes a terminated
//     wave out of the barrier's count, HIP leaves a barrier that not every thread reaches undefined;
//   - values that cross workgroups inside a launch are published with relaxed agent-scope atomics ordered by s_waitcnt
//     vmcnt(0) (publish / observe / published, last_workgroup_of_problem): sound because an sc1 store is acknowledged at
//     the memory-side coherence point on gfx942 / gfx950, a data race under the HSA memory model;
//   - the DPP reductions spell out the wait states the hazard recogniser would insert (acmpc_device.h).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "acmpc kernels are written for gfx950 (MI355X): see the note above before building for another architecture"
#endif

namespace acmpc {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Phase stamps for tools/archive/solo_probe.hip (a standalone build of this file with -DACMPC_STAMPS); nothing in the library.
#ifdef ACMPC_STAMPS
__device__ unsigned long long g_stamps[4096 * 16];
#define ACMPC_STAMP(slot)                                                                                         \
  do {                                                                                                            \
    if ((threadIdx.x & 63) == 0)                                                                                  \
      g_stamps[((blockIdx.y * gridDim.x + blockIdx.x) * 2 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define ACMPC_STAMP(slot) \
  do {                    \
  } while (0)
#endif

template <int CPT>
struct VecOf;
template <>
struct VecOf<1> {
  using type = float;
};
template <>
struct VecOf<2> {
  using type = f32x2;
};
template <>
struct VecOf<4> {
  using type = f32x4;
};

template <int CPT>
__device__ __forceinline__ void unpack(const typename VecOf<CPT>::type& v, float (&out)[CPT]) {
  if constexpr (CPT == 1) {
    out[0] = v;
  } else {
#pragma unroll
    for (int j = 0; j < CPT; ++j) out[j] = v[j];
  }
}

// Controls of CPT adjacent candidates at step i.
template <int LAYOUT, int CPT>
__device__ __forceinline__ void load_controls(const float* __restrict__ U, int p, int N, int n, int i, int c0,
                                              float (&v)[CPT], float (&k)[CPT]) {
  if constexpr (LAYOUT == 1) {
    // U[p][i][0|1][c]: lanes read consecutive candidates -> one fully coalesced wave access per component
    using V = typename VecOf<CPT>::type;
    const float* row = U + (static_cast<size_t>(p) * n + i) * 2 * static_cast<size_t>(N) + c0;
    unpack<CPT>(__builtin_nontemporal_load(reinterpret_cast<const V*>(row)), v);
    unpack<CPT>(__builtin_nontemporal_load(reinterpret_cast<const V*>(row + N)), k);
  } else {
    // U[p][c][i][0|1]: 8-byte (v, kappa) pairs at a row stride of 8n bytes
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const f32x2 vk = *reinterpret_cast<const f32x2*>(U + ((static_cast<size_t>(p) * N + c0 + j) * n + i) * 2);
      v[j] = vk[0];
      k[j] = vk[1];
    }
  }
}

// Publishing between workgroups of ONE launch without fences.  An agent-scope fence is an L2 write-back (release) or an
// L2 invalidate (acquire) on this multi-die part - microseconds each, and the fused finalize needed three per
// workgroup.  Instead the few values that cross workgroups (partial keys, traces, tickets) are written and read with
// agent-scope atomic stores / loads, which go to the memory-side coherence point past the per-die L2, and a writer
// only has to wait until its stores have been acknowledged (vmcnt = 0) before it takes its ticket.
template <typename T>
__device__ __forceinline__ void publish(T* where, T value) {
  __hip_atomic_store(where, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T observe(const T* where) {
  return __hip_atomic_load(where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void published() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every store of this wave has been acknowledged
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// Candidate 2 of a sampled round is the LQ plan (SampleArgs::u_extra) - ONE candidate of the launch, in one workgroup.
// The plan is read in place from pinned HOST memory (the tick's host computes it while the first round runs): only the
// workgroup that holds global index 2 fetches it.  With every workgroup staging it the last round of a tick moved
// 256 x 392 B over PCIe for one lane's sake - two microseconds of its fourteen.
__device__ __forceinline__ bool holds_candidate_2(const RolloutArgs& a, const SampleArgs& smp) {
  const int64_t first = a.index_offset + static_cast<int64_t>(blockIdx.x) * kWave;
  return smp.u_extra != nullptr && first <= 2 && 2 < first + kWave;
}

// Phase stamps of the mode T rollout for tools/modeT_stamps.py (a scratch build of the library with -DACMPC_T_STAMPS,
// tools/ab_build.sh): lane 0 of every wave stamps the 100 MHz wall clock at entry, after the tables are staged, after the
// step loop and at its end, and leaves its place on the chip (XCC_ID, HW_ID) beside them.  Nothing in the library.
#ifdef ACMPC_T_STAMPS
constexpr int kStampWaves = 1 << 17;
__device__ unsigned long long g_t_stamps[kStampWaves * 6];
#define ACMPC_T_STAMP(slot)                                                                                      \
  do {                                                                                                           \
    if constexpr (MODE == 1) {                                                                                   \
      const unsigned wave_ = (blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / kWave) + (threadIdx.x / kWave);   \
      if ((threadIdx.x & (kWave - 1)) == 0 && wave_ < kStampWaves) {                                             \
        g_t_stamps[wave_ * 6 + (slot)] = wall_clock64();                                                         \
        if ((slot) == 0) {                                                                                       \
          g_t_stamps[wave_ * 6 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   /* HW_ID */         \
          g_t_stamps[wave_ * 6 + 5] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   /* XCC_ID */        \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
#else
#define ACMPC_T_STAMP(slot) \
  do {                      \
  } while (0)
#endif

// PACK = candidates per arithmetic state: 2 = pairs in v_pk_* instructions, 1 = plain float32 instructions.
// Mode T with two candidates per lane needs 67 VGPRs as the compiler allocates it freely: seven waves per SIMD, where a
// launch of 1 M candidates is eight - the eighth workgroup of every CU then runs alone after the others (a second
// generation of lone waves: +15 % on the launch).  Asking for eight waves per SIMD caps the allocation at 64.
// WAVES = 8 asks for that many waves per SIMD, which caps the allocation at 64 VGPRs.  Measured, 1 M candidates (256 poses
// x 4 096), same box, 67 VGPRs / capped: verified 16-waypoint search 331 / 283 us, 4-waypoint window 102.5 / 92.8 us -
// but the 8-waypoint window 134.9 / 141.3 us (its waves already queue for the LDS: an eighth wave per SIMD adds to the
// queue what it saves on the tail), so the launcher caps every search but that one.  (The verified search, since round 3
// an 8-waypoint window + its certificate: 190 us capped, 225 us uncapped.)
template <int MODE, int LAYOUT, int CPT, int BLOCK, int PACK, bool PUBLISH>
__device__ __forceinline__ void rollout_block(const RolloutArgs& a, unsigned char* smem, const int p) {
  // carve: [0,32) wave keys | [32,48) wave feasible counts | [64, ...) mode-T waypoint table
  int64_t* s_key = reinterpret_cast<int64_t*>(smem);
  int* s_feas = reinterpret_cast<int*>(smem + 32);
  float* s_wp = reinterpret_cast<float*>(smem + 64);

  const int tid = threadIdx.x;
  const int c0 = (blockIdx.x * BLOCK + tid) * CPT;
  const bool active = c0 < a.N;  // N % CPT == 0 is guaranteed by the launcher
  const int n = a.n;
  const Weights w = a.w;
  constexpr int kStride = (MODE == 0) ? kCoefS : kCoefT;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kStride;
  const float* __restrict__ x0 = a.x0 + p * 3;

  ACMPC_T_STAMP(0);
  if (a.start_clock != nullptr && threadIdx.x == 0)   // (wave-uniform: a scalar compare when the diagnostic is off)
    a.start_clock[static_cast<size_t>(p) * gridDim.x + blockIdx.x] = wall_clock64();
  float* s_xy = s_wp + n * kCoefT;  // the nearest-waypoint search's key table: (a, b, c) per waypoint (search_entry)
  float* s_frames = s_xy + ((kKeyStride * n + 3) & ~3);  // frames of the verified search (exhaustive semantics), when given
  if constexpr (MODE == 1) {
    stage_temporal_tables(coef, n, tid, BLOCK, s_wp, s_xy);
    if (a.nn_frames != nullptr) {
      const float* __restrict__ frames = a.nn_frames + static_cast<size_t>(p) * verified_frame_floats(n);
      for (int e = tid; e < verified_frame_floats(n); e += BLOCK) s_frames[e] = frames[e];
    }
    __syncthreads();
  }
  ACMPC_T_STAMP(1);

  float cost[CPT];
  bool feas[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    cost[j] = __builtin_inff();
    feas[j] = false;
  }

  // Mode T's verified nearest-waypoint search has a wave-cooperative fallback that every lane must reach, so there
  // the tail lanes of the last workgroup roll a valid dummy (the problem's last candidates) instead of idling.
  const bool run = active || (MODE == 1 && a.nn_frames != nullptr);
  const int c_run = active ? c0 : max(a.N - CPT, 0);
  if (run) {
    constexpr int kPack = PACK;
    static_assert(CPT % PACK == 0, "a lane's candidates split evenly into arithmetic states");
    constexpr int kGroups = CPT / kPack;
    using F = typename std::conditional<kPack == 2, f32x2, float>::type;
    using I = typename IndexOf<F>::type;
    auto pack = [](const float (&src)[CPT], int g) {
      if constexpr (kPack == 2) {
        F out;
        out[0] = src[2 * g];
        out[1] = src[2 * g + 1];
        return out;
      } else {
        return src[g];
      }
    };
    auto unpack_to = [](F value, float (&dst)[CPT], int g) {
      if constexpr (kPack == 2) {
        dst[2 * g] = value[0];
        dst[2 * g + 1] = value[1];
      } else {
        dst[g] = value;
      }
    };
    float viol[CPT];
    if constexpr (MODE == 0) {
      StateS_<F> st[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
        st[g] = StateS_<F>{splat<F>(x0[0]), splat<F>(x0[1]), splat<F>(x0[2]), splat<F>(0.0f), splat<F>(0.0f)};
#pragma unroll 7
      for (int i = 0; i < n; ++i) {
        float v[CPT], k[CPT];
        load_controls<LAYOUT, CPT>(a.U, p, a.N, n, i, c_run, v, k);
        const float* __restrict__ c = coef + i * kCoefS;  // wave-uniform -> scalar loads
#pragma unroll
        for (int g = 0; g < kGroups; ++g) step_spatial<F>(st[g], c, pack(v, g), pack(k, g), w);
      }
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        unpack_to(finish_spatial<F>(st[g], w), cost, g);
        unpack_to(st[g].V, viol, g);
      }
    } else {
      StateT_<F> st[kGroups];
      I nearest[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        st[g] = start_temporal<F>(x0, coef);
        nearest[g] = I(0);
      }
      with_search_kind(w, n, [&](auto kind) {
        for (int i = 0; i < n; ++i) {
          // A launch of ONE generation (a.even_progress, set by the launcher): the hardware issues from the oldest wave
          // first, so the eight waves of a SIMD finish one after the other - the first after half the launch, the last
          // alone, with nothing to hide its latencies behind - and the launch ends a quarter later than the SIMD's
          // instructions take.  A wave that is ahead yields instead: priority 3 in the first quarter of the horizon down to
          // 0 in the last; the waves stay within a quarter of each other and leave together (1 M candidates, one box:
          // resident share of the launch 0.57-0.73 -> 0.89, 110 -> 98 us; tools/modeT_stamps.py).  With several
          // generations the staggered ends are what overlaps a new workgroup's staging with its neighbours' arithmetic:
          // there the flag stays off (16.8 M: 1 % slower with it).
          if (a.even_progress != 0 && (i & 3) == 0) {
            const int quarter = (4 * i) / n;
            if (quarter == 0) __builtin_amdgcn_s_setprio(3);
            else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
            else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
          float v[CPT], k[CPT];
          load_controls<LAYOUT, CPT>(a.U, p, a.N, n, i, c_run, v, k);
          if constexpr (decltype(kind)::value == kSearchVerified) {
            // phases across the lane's candidates: advance + window search of all (straight-line code), then the
            // wave-wide fallback for whatever was not certified, then rows and costs
            int uncertified[kGroups];
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
              temporal_advance<F>(st[g], pack(v, g), pack(k, g), w);
              nearest[g] = verified_window(st[g], s_xy, s_frames, n, nearest[g], uncertified[g]);
            }
            int any = 0;
#pragma unroll
            for (int g = 0; g < kGroups; ++g) any |= uncertified[g];
            if (__ballot(any != 0) != 0ull) {
#pragma unroll
              for (int g = 0; g < kGroups; ++g) nearest[g] = verified_fix(st[g], s_xy, n, nearest[g], uncertified[g]);
            }
#pragma unroll
            for (int g = 0; g < kGroups; ++g) temporal_settle(st[g], s_wp, nearest[g], pack(v, g), pack(k, g), w);
          } else {
#pragma unroll
            for (int g = 0; g < kGroups; ++g)
              nearest[g] = step_temporal_as<decltype(kind)::value>(st[g], s_wp, s_xy, n, pack(v, g), pack(k, g), w,
                                                                   nearest[g], s_frames);
          }
        }
      }, a.nn_frames != nullptr);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        unpack_to(finish_temporal<F>(st[g], n, w), cost, g);
        unpack_to(st[g].V, viol, g);
      }
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) feas[j] = viol[j] == 0.0f;
    if (a.costs != nullptr && active) {
      using V = typename VecOf<CPT>::type;
      float* out = a.costs + static_cast<size_t>(p) * a.N + c0;
      if constexpr (CPT == 1) {
        out[0] = cost[0];
      } else {
        V packed;
#pragma unroll
        for (int j = 0; j < CPT; ++j) packed[j] = cost[j];
        *reinterpret_cast<V*>(out) = packed;
      }
    }
  }

  ACMPC_T_STAMP(2);
  // (cost, index) argmin: thread -> wave (shuffles) -> workgroup (LDS) -> one partial per workgroup
  int64_t key = kKeyMax;
  int nfeas = 0;
  if (active) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int64_t kj = pack_key(cost[j], static_cast<uint32_t>(a.index_offset + c0 + j));
      key = (kj < key) ? kj : key;
      nfeas += feas[j] ? 1 : 0;
    }
  }
  key = wave_min_key(key);
  nfeas = wave_sum_int(nfeas);
  constexpr int kWaves = BLOCK / kWave;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  if constexpr (kWaves > 1) {
    if (lane == 0) {
      s_key[wave] = key;
      s_feas[wave] = nfeas;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int q = 1; q < kWaves; ++q) {
        key = (s_key[q] < key) ? s_key[q] : key;
        nfeas += s_feas[q];
      }
    }
  }
  if (tid == 0) {
    const size_t slot = static_cast<size_t>(p) * gridDim.x + blockIdx.x;
    if constexpr (PUBLISH) {   // read by another workgroup of THIS launch (rollout_tailed_kernel): to the coherence point
      publish(&a.partial_keys[slot], key);
      publish(&a.partial_feas[slot], nfeas);
    } else {
      a.partial_keys[slot] = key;
      a.partial_feas[slot] = nfeas;
    }
  }
  ACMPC_T_STAMP(3);
}

template <int MODE, int LAYOUT, int CPT, int BLOCK, int PACK = (CPT >= 2 ? 2 : 1), int WAVES = 1>
__global__ void __launch_bounds__(BLOCK, WAVES) rollout_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  rollout_block<MODE, LAYOUT, CPT, BLOCK, PACK, false>(a, smem, static_cast<int>(blockIdx.y));
}

// Candidate-major control matrix U[P][N][n][2] (what NumPy host code holds): a wave's 64 candidates are 64
// consecutive rows = ONE contiguous span of 64 * 8n bytes.  The wave copies that span into LDS with 16-byte loads
// (every HBM line fetched exactly once, fully coalesced) and then walks the steps reading its own row with
// ds_read_b64: the row pitch is 2n dwords, which for odd n (every horizon the reference uses) lands the 32 lanes
// of a read group on 32 distinct bank pairs - conflict-free without padding.  One wave per workgroup, so the LDS
// budget (8n * 64 bytes = 25 KB at H = 50) sets the occupancy: 6 waves per CU, each with its whole tile in flight.
// Measured 3.2 TB/s at H = 50 (a chunked, software-pipelined variant with 16 waves per CU and 8-byte row-wise loads
// measured 2.9 TB/s, plain per-lane strided loads 3.0 TB/s): the step-major layout is the fast path.
template <int MODE>
__global__ void __launch_bounds__(kWave) rollout_tile_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_tile[];  // [64][2n] then (mode T) the waypoint table
  const int p = blockIdx.y;
  const int lane = threadIdx.x;
  const int c0 = blockIdx.x * kWave;
  const int rows = min(kWave, a.N - c0);
  const int n = a.n;
  const int row_floats = 2 * n;
  const Weights w = a.w;
  constexpr int kStride = (MODE == 0) ? kCoefS : kCoefT;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kStride;
  const float* __restrict__ x0 = a.x0 + p * 3;
  float* s_wp = s_tile + ((kWave * row_floats + 3) & ~3);

  const size_t first = (static_cast<size_t>(p) * a.N + c0) * row_floats;  // float index of the span
  const float* __restrict__ src = a.U + first;
  const int total = rows * row_floats;
  if ((first & 3) == 0) {
    const f32x4* __restrict__ src4 = reinterpret_cast<const f32x4*>(src);
    f32x4* dst4 = reinterpret_cast<f32x4*>(s_tile);
    const int quads = total >> 2;
#pragma unroll 8
    for (int q = lane; q < quads; q += kWave) dst4[q] = __builtin_nontemporal_load(src4 + q);
    for (int e = (quads << 2) + lane; e < total; e += kWave) s_tile[e] = src[e];
  } else {  // span starts on an 8-byte boundary only (odd p * N): 8-byte copies
    const f32x2* __restrict__ src2 = reinterpret_cast<const f32x2*>(src);
    f32x2* dst2 = reinterpret_cast<f32x2*>(s_tile);
#pragma unroll 8
    for (int q = lane; q < (total >> 1); q += kWave) dst2[q] = __builtin_nontemporal_load(src2 + q);
  }
  float* s_xy = s_wp + n * kCoefT;
  if constexpr (MODE == 1) {
    stage_temporal_tables(coef, n, lane, kWave, s_wp, s_xy);
  }
  __syncthreads();

  const bool active = lane < rows;
  float cost = __builtin_inff();
  bool feas = false;
  if (active) {
    const f32x2* row = reinterpret_cast<const f32x2*>(s_tile + lane * row_floats);
    if constexpr (MODE == 0) {
      StateS st{x0[0], x0[1], x0[2], 0.0f, 0.0f};
#pragma unroll 7
      for (int i = 0; i < n; ++i) {
        const f32x2 vk = row[i];
        step_spatial(st, coef + i * kCoefS, vk[0], vk[1], w);
      }
      cost = finish_spatial(st, w);
      feas = st.V == 0.0f;
    } else {
      StateT st = start_temporal<float>(x0, coef);
      int nearest = 0;
      for (int i = 0; i < n; ++i) {
        const f32x2 vk = row[i];
        nearest = step_temporal(st, s_wp, s_xy, n, vk[0], vk[1], w, nearest);
      }
      cost = finish_temporal(st, n, w);
      feas = st.V == 0.0f;
    }
    if (a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c0 + lane] = cost;
  }
  int64_t key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c0 + lane)) : kKeyMax;
  int nfeas = (active && feas) ? 1 : 0;
  key = wave_min_key(key);
  nfeas = wave_sum_int(nfeas);
  if (lane == 0) {
    const size_t slot = static_cast<size_t>(p) * gridDim.x + blockIdx.x;
    a.partial_keys[slot] = key;
    a.partial_feas[slot] = nfeas;
  }
}

// ---- candidate sampling ------------------------------------------------------------------------------------
// U_c = clip(centre + a_c * sigma * (smooth noise)), one lane per candidate.  The noise is a raised-cosine blend of
// kSampleKnots x 2 standard normals per candidate (smooth along the horizon), a_c cycles through 8 amplitude
// levels, candidate 0 is the centre itself (so a round can never lose the incumbent) and candidate 1 the reference
// controls.  Philox counters are (global candidate, problem, round, draw): reproducible on any rank.
template <int LAYOUT>
__global__ void __launch_bounds__(256) sample_kernel(const SampleArgs a) {
  const int p = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int n = a.n;
  const SampleSpec sp = a.spec;
  const uint32_t gidx = static_cast<uint32_t>(a.index_offset + c);
  float z[kKnots][2];
  draw_normals(sp, gidx, static_cast<uint32_t>(p), z);
  // candidate 1 = the reference controls, candidate 2 = `u_extra` (the LQ plan), each when given: amplitude 0, own centre
  const float* alt = (gidx == 1u) ? a.u_ref : (gidx == 2u) ? a.u_extra : nullptr;
  const bool use_ref = alt != nullptr;
  const float amp = use_ref ? 0.0f : candidate_amplitude(gidx);
  const float* __restrict__ centre =
      use_ref ? alt + static_cast<size_t>(p) * n * 2 : a.centre + static_cast<size_t>(p) * a.centre_stride;
  // blend weights and the centre sequence -> LDS once per workgroup; the knot boundaries are kernel arguments, so
  // neither the loop bounds nor the per-step operands wait on a dependent scalar/global load (those dependencies used
  // to make this the longest kernel of an optimisation round)
  extern __shared__ __attribute__((aligned(16))) float s_smp[];  // [n] weights, then [n][2] centre
  float* s_w0 = s_smp;
  float* s_centre = s_smp + ((n + 3) & ~3);
  for (int e = threadIdx.x; e < n; e += 256) s_w0[e] = sp.segments[2 * e + 1];
  // candidate 1 of a workgroup that holds it reads u_ref instead; every other lane the centre
  const float* __restrict__ block_centre = a.centre + static_cast<size_t>(p) * a.centre_stride;
  for (int e = threadIdx.x; e < 2 * n; e += 256) s_centre[e] = block_centre[e];
  __syncthreads();
  if (c >= a.N) return;
#pragma unroll
  for (int knot = 0; knot < kKnots - 1; ++knot) {
    for (int i = sp.knot_begin[knot]; i < sp.knot_begin[knot + 1]; ++i) {
      float v, k;
      const float cv = use_ref ? centre[2 * i] : s_centre[2 * i];
      const float ck = use_ref ? centre[2 * i + 1] : s_centre[2 * i + 1];
      blend_control(sp, amp, s_w0[i], cv, ck, z[knot][0], z[knot][1], z[knot + 1][0], z[knot + 1][1], v, k);
      if constexpr (LAYOUT == 1) {
        float* row = a.U + (static_cast<size_t>(p) * n + i) * 2 * static_cast<size_t>(a.N) + c;
        row[0] = v;
        row[a.N] = k;
      } else {
        f32x2 vk;
        vk[0] = v;
        vk[1] = k;
        *reinterpret_cast<f32x2*>(a.U + ((static_cast<size_t>(p) * a.N + c) * n + i) * 2) = vk;
      }
    }
  }
}

// sample_kernel and rollout_kernel fused for the latency-bound closed-loop solve: one lane = one candidate whose
// controls are drawn (same Philox counters, same blend_control arithmetic: bit-identical to sample_kernel) and
// consumed step by step without ever being written to memory.  Output: the per-workgroup partial keys, as usual.
// One wave per workgroup and one workgroup per CU at the closed-loop size (16 384 candidates), so nothing hides a
// wait: what the kernel does about each is said where it is done (operands through LDS, requests before the draws).
//
// `traced`: every lane also leaves what it computed - controls and state of each step, violation, cost - in its column
// of an LDS block [5n + 2][64] at `trace_lds_floats`, and the workgroup copies the column of its best candidate to
// `trace_out` (global, [5n + 2]): the fused finalize then assembles the winner's record out of the winning workgroup's
// trace instead of drawing and rolling that candidate a second time, and a following round finds its centre there.
// Rows: 2i, 2i + 1 = (v, kappa) of step i | 2n + 3i .. + 2 = state after step i | 5n = V | 5n + 1 = cost.
// (The trace is addressed off `s_fused` itself, not through a pointer that may be null: a select of pointers would lose
// the LDS address space.)
constexpr int kStagedSteps = 128;   // steps whose uniform operands travel through registers (longer horizons: the rest by a loop)

template <int MODE>
__device__ __forceinline__ void rollout_sampled_body(const RolloutArgs& a, const SampleArgs& smp, float* s_fused,
                                                     const int uniform_lds_floats, const bool traced = false,
                                                     const int trace_lds_floats = 0, float* trace_out = nullptr) {
  float* s_trace = s_fused + trace_lds_floats;
  const int p = blockIdx.y;
  const int lane = threadIdx.x;
  const int c = blockIdx.x * kWave + lane;
  const int n = a.n;
  const Weights w = a.w;
  const SampleSpec sp = smp.spec;
  constexpr int kStride = (MODE == 0) ? kCoefS : kCoefT;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kStride;
  const float* __restrict__ x0 = a.x0 + p * 3;
  const float* __restrict__ centre = smp.centre + static_cast<size_t>(p) * smp.centre_stride;
  const float* __restrict__ ref = (smp.u_ref != nullptr) ? smp.u_ref + static_cast<size_t>(p) * n * 2 : nullptr;
  const float* __restrict__ extra = holds_candidate_2(a, smp) ? smp.u_extra + static_cast<size_t>(p) * n * 2 : nullptr;
  const float* __restrict__ knot_weight = sp.segments;
  float* s_wp = s_fused;
  float* s_xy = s_wp + n * kCoefT;
  if constexpr (MODE == 1) {
    stage_temporal_tables(coef, n, lane, kWave, s_wp, s_xy);
    __syncthreads();
  }

  // The wave-uniform operands of the steps - table rows, centre and reference controls, knot weights: 17 floats per
  // step - go through LDS.  Read where they are needed they are scalar loads that miss this CU's cold scalar cache
  // (58 lines at H = 50), and a lone wave has nothing to hide a miss behind: half of this kernel's wave-cycles were
  // such waits.  Here the lanes fetch them side by side (one round trip, covered by the Philox draws below), and the
  // step loop reads them back as broadcast LDS reads one step ahead of the arithmetic.
  //   s_uni: [n][12] table rows (mode S) | [n][2] centre | [n][2] reference (or the centre again) | [n] knot weights
  float* s_uni = s_fused + uniform_lds_floats;
  float* s_row = s_uni;
  float* s_centre = s_row + ((MODE == 0) ? n * kCoefS : 0);
  float* s_ref = s_centre + 2 * n;
  float* s_extra = s_ref + 2 * n;    // candidate 2's controls (smp.u_extra: the LQ plan), when given
  float* s_weight = s_extra + 2 * n;
  const bool chained = smp.prev_keys != nullptr;   // wave-uniform
  constexpr int kRowQuads = (kStagedSteps * kCoefS / 4 + kWave - 1) / kWave;   // table quads per lane
  constexpr int kPairs = (kStagedSteps + kWave - 1) / kWave;                    // (v, kappa) pairs per lane
  f32x4 g_row[kRowQuads];
  f32x2 g_centre[kPairs], g_ref[kPairs], g_extra[kPairs], g_weight[kPairs];
#pragma unroll
  for (int q = 0; q < kRowQuads; ++q) {
    if constexpr (MODE == 0)
      g_row[q] = reinterpret_cast<const f32x4*>(coef)[min(lane + q * kWave, n * (kCoefS / 4) - 1)];
  }
#pragma unroll
  for (int q = 0; q < kPairs; ++q) {
    const int j = min(lane + q * kWave, n - 1);
    if (!chained) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[j];
    if (ref != nullptr) g_ref[q] = reinterpret_cast<const f32x2*>(ref)[j];
    if (extra != nullptr) g_extra[q] = reinterpret_cast<const f32x2*>(extra)[j];
    g_weight[q] = reinterpret_cast<const f32x2*>(knot_weight)[j];   // (knot, weight) pairs: the weight is [1]
  }
  // chained rounds: the previous launch's partial keys, four per lane, requested with the rest
  int64_t prev_key[kChainBlocks / kWave];
  if (chained) {
#pragma unroll
    for (int q = 0; q < kChainBlocks / kWave; ++q) {
      const int b = lane + q * kWave;
      prev_key[q] = smp.prev_keys[static_cast<size_t>(p) * smp.prev_blocks + min(b, smp.prev_blocks - 1)];
    }
  }

  const bool active = c < a.N;
  float cost = __builtin_inff();
  bool feas = false;
  float z[kKnots][2] = {};
  const uint32_t gidx = static_cast<uint32_t>(a.index_offset + c);
  // first half of the draws (covers the round trip of the requests above), then - chained - the winner of the previous
  // round and the request for ITS controls, which the second half of the draws covers
  if (active) draw_normals<0, kKnots / 4>(sp, gidx, static_cast<uint32_t>(p), z);
  if (chained) {   // the previous round's winner: argmin over its workgroups' keys; its controls head that workgroup's trace
    int64_t best = kKeyMax;
#pragma unroll
    for (int q = 0; q < kChainBlocks / kWave; ++q) {
      const int64_t kb = (lane + q * kWave < smp.prev_blocks) ? prev_key[q] : kKeyMax;
      best = (kb < best) ? kb : best;
    }
    // (workgroup b rolled the candidates b * 64 ..: the winner's workgroup follows from its index)
    const int64_t winner = wave_min_key(best);
    const int block = static_cast<int>(static_cast<int64_t>(static_cast<uint32_t>(winner & 0xffffffffLL)) - a.index_offset) / kWave;
    centre = smp.prev_trace + (static_cast<size_t>(p) * smp.prev_blocks + block) * smp.prev_pitch;
#pragma unroll
    for (int q = 0; q < kPairs; ++q) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[min(lane + q * kWave, n - 1)];
  }
  if (active) draw_normals<kKnots / 4, kKnots / 2>(sp, gidx, static_cast<uint32_t>(p), z);
#pragma unroll
  for (int q = 0; q < kRowQuads; ++q) {
    if constexpr (MODE == 0) {
      const int e = lane + q * kWave;
      if (e < n * (kCoefS / 4)) reinterpret_cast<f32x4*>(s_row)[e] = g_row[q];
    }
  }
#pragma unroll
  for (int q = 0; q < kPairs; ++q) {
    const int j = lane + q * kWave;
    if (j < n) {
      reinterpret_cast<f32x2*>(s_centre)[j] = g_centre[q];
      reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? g_ref[q] : g_centre[q];
      if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = g_extra[q];
      s_weight[j] = g_weight[q][1];
    }
  }
  if (n > kStagedSteps) {   // horizons beyond the registers' share: plain copies (a second round trip)
    if constexpr (MODE == 0)
      for (int e = lane + kRowQuads * kWave; e < n * (kCoefS / 4); e += kWave)
        reinterpret_cast<f32x4*>(s_row)[e] = reinterpret_cast<const f32x4*>(coef)[e];
    for (int j = lane + kPairs * kWave; j < n; j += kWave) {
      const f32x2 cj = reinterpret_cast<const f32x2*>(centre)[j];
      reinterpret_cast<f32x2*>(s_centre)[j] = cj;
      reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? reinterpret_cast<const f32x2*>(ref)[j] : cj;
      if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = reinterpret_cast<const f32x2*>(extra)[j];
      s_weight[j] = knot_weight[2 * j + 1];
    }
  }
  __syncthreads();   // one wave: orders the staging above before the reads below
  if (active) {
    // candidate 1 = the reference controls, candidate 2 = `u_extra` (the LQ plan), each when given: amplitude 0, own centre
    const bool use_extra = (gidx == 2u) && (smp.u_extra != nullptr);
    const bool use_ref = ((gidx == 1u) && (smp.u_ref != nullptr)) || use_extra;
    const f32x2* s_alt = reinterpret_cast<const f32x2*>(use_extra ? s_extra : s_ref);   // (per lane)
    const float amp = use_ref ? 0.0f : candidate_amplitude(gidx);
    StateS ss{x0[0], x0[1], x0[2], 0.0f, 0.0f};
    StateT ts = start_temporal<float>(x0, coef);
    int nearest = 0;
    struct StepOperands {
      f32x2 centre, ref;
      float weight;
      f32x4 lo, hi;   // table row (mode S): ds, a21, a31, b31 | f3, v_ref, k_ref, ey_lo
      float last;     // ey_hi
    };
    // (the request behind the last step reads inside the block - the next array, or the padding - and is never used)
    auto request = [&](int i) {
      StepOperands o;
      o.centre = reinterpret_cast<const f32x2*>(s_centre)[i];
      o.ref = s_alt[i];
      o.weight = s_weight[i];
      if constexpr (MODE == 0) {
        o.lo = *reinterpret_cast<const f32x4*>(s_row + i * kCoefS);
        o.hi = *reinterpret_cast<const f32x4*>(s_row + i * kCoefS + 4);
        o.last = s_row[i * kCoefS + 8];
      }
      return o;
    };
    StepOperands now = request(sp.knot_begin[0]);
#pragma unroll
    for (int knot = 0; knot < kKnots - 1; ++knot) {
#pragma unroll 2
      for (int i = sp.knot_begin[knot]; i < sp.knot_begin[knot + 1]; ++i) {
        const StepOperands next = request(i + 1);
        float v, k;
        const float cv = use_ref ? now.ref[0] : now.centre[0];
        const float ck = use_ref ? now.ref[1] : now.centre[1];
        blend_control(sp, amp, now.weight, cv, ck, z[knot][0], z[knot][1], z[knot + 1][0], z[knot + 1][1], v, k);
        if constexpr (MODE == 0) {
          const float row[9] = {now.lo[0], now.lo[1], now.lo[2], now.lo[3], now.hi[0],
                                now.hi[1], now.hi[2], now.hi[3], now.last};
          step_spatial<float>(ss, row, v, k, w);
        } else {
          nearest = step_temporal(ts, s_wp, s_xy, n, v, k, w, nearest);
        }
        now = next;
        if (traced) {   // wave-uniform
          float* col = s_trace + lane;
          col[(2 * i) * kWave] = v;
          col[(2 * i + 1) * kWave] = k;
          col[(2 * n + 3 * i) * kWave] = (MODE == 0) ? ss.ey : ts.X + coef[0];       // (poses leave in the caller's frame)
          col[(2 * n + 3 * i + 1) * kWave] = (MODE == 0) ? ss.ep : ts.Y + coef[1];
          col[(2 * n + 3 * i + 2) * kWave] = (MODE == 0) ? ss.t : ts.phi;
        }
      }
    }
    if constexpr (MODE == 0) {
      cost = finish_spatial<float>(ss, w);
      feas = ss.V == 0.0f;
    } else {
      cost = finish_temporal<float>(ts, n, w);
      feas = ts.V == 0.0f;
    }
    if (traced) {
      s_trace[(5 * n) * kWave + lane] = (MODE == 0) ? ss.V : ts.V;
      s_trace[(5 * n + 1) * kWave + lane] = cost;
    }
    if (a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c] = cost;
  }
  const int64_t own_key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c)) : kKeyMax;
  int nfeas = (active && feas) ? 1 : 0;
  int best_lane;
  const int64_t key = wave_min_key_by_lane(own_key, best_lane);   // (the index rises with the lane)
  nfeas = wave_sum_int(nfeas);
  if (traced) {
    __syncthreads();   // one wave: orders the column writes above before the row reads
    for (int e = lane; e < 5 * n + 2; e += kWave) publish(&trace_out[e], s_trace[e * kWave + best_lane]);
  }
  if (lane == 0) {
    const size_t slot = static_cast<size_t>(p) * gridDim.x + blockIdx.x;
    publish(&a.partial_keys[slot], key);
    publish(&a.partial_feas[slot], nfeas);
  }
}

// The same candidate's control at ONE step, for the finalize kernel's lane-per-step regeneration: identical
// arithmetic (blend_control on the same operands), the bracketing knots picked by a select chain.
__device__ __forceinline__ void regenerate_control(const SampleSpec& sp, const float (&z)[kKnots][2], float amp,
                                                   const float* centre, int i, float& v, float& k) {
  const int k0 = static_cast<int>(sp.segments[2 * i]);
  float z0v = z[0][0], z0k = z[0][1], z1v = z[1][0], z1k = z[1][1];
#pragma unroll
  for (int knot = 1; knot < kKnots - 1; ++knot) {
    const bool hit = (k0 == knot);
    z0v = hit ? z[knot][0] : z0v;
    z0k = hit ? z[knot][1] : z0k;
    z1v = hit ? z[knot + 1][0] : z1v;
    z1k = hit ? z[knot + 1][1] : z1k;
  }
  blend_control(sp, amp, sp.segments[2 * i + 1], centre[2 * i], centre[2 * i + 1], z0v, z0k, z1v, z1k, v, k);
}

// One wave per problem.  The winner is re-rolled by the whole wave in lock-step: lane l fetches step l's controls
// and table row (one round of parallel loads), each step's inputs are then broadcast with v_readlane (SGPRs), so
// the sequential chain touches no memory at all; every lane carries the same state and the record image is
// assembled in LDS and written out by all lanes.  Same step functions as the rollout kernel -> same bits.
__device__ __forceinline__ float bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float key_cost(int64_t key) {
  const int32_t hi = static_cast<int32_t>(key >> 32);
  union {
    int32_t i;
    float f;
  } b;
  b.i = (hi >= 0) ? hi : (hi ^ 0x7fffffff);
  return b.f;
}

// `s_rec`: LDS for the record image [4 + 2n + 3(n+1)] and, in mode T, the waypoint table behind it.  Called by one
// whole wave (the only wave of its workgroup).  The partial keys are read with agent-scope atomic loads: when the
// caller is the last workgroup of a fused rollout (below) they were written by other workgroups of the SAME launch.
template <int MODE, int LAYOUT>
__device__ __forceinline__ void finalize_problem(const FinalizeArgs& a, const int p, float* s_rec) {
  const int lane = threadIdx.x;
  const int n = a.n;

  // (mode S) lane l's table row of the first 64 steps and the start state do not depend on who won: requested first,
  // they travel together with the partial keys instead of after the argmin
  float row_first[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  float x0_first[3] = {0.0f, 0.0f, 0.0f};
  if constexpr (MODE == 0) {
    if (a.records != nullptr) {
      const float* __restrict__ coef_p = a.coef + static_cast<size_t>(p) * n * kCoefS;
#pragma unroll
      for (int q = 0; q < 9; ++q) row_first[q] = coef_p[min(lane, n - 1) * kCoefS + q];
#pragma unroll
      for (int q = 0; q < 3; ++q) x0_first[q] = a.x0[p * 3 + q];
    }
  }

  int nfeas = 0;
  int64_t key = kKeyMax;
  for (int b = lane; b < a.blocks_per_problem; b += kWave) {
    const size_t slot = static_cast<size_t>(p) * a.blocks_per_problem + b;
    nfeas += __hip_atomic_load(&a.partial_feas[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t kb = __hip_atomic_load(&a.partial_keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    key = (kb < key) ? kb : key;
  }
  nfeas = wave_sum_int(nfeas);
  key = wave_min_key(key);
  if (a.keys_in != nullptr) key = a.keys_in[p];
  if (a.keys_out != nullptr && lane == 0) a.keys_out[p] = key;
  if (a.records == nullptr) return;

  const int rec_floats = 4 + 2 * n + 3 * (n + 1);
  float* __restrict__ rec = a.records + static_cast<size_t>(p) * rec_floats;
  const int64_t local = static_cast<int64_t>(static_cast<uint32_t>(key & 0xffffffffLL)) - a.index_offset;
  const bool owner = a.regenerate || (local >= 0 && local < a.N);  // wave-uniform
  if (!owner) {
    for (int e = lane; e < rec_floats; e += kWave) rec[e] = (e == 2) ? static_cast<float>(nfeas) : 0.0f;
    return;
  }

  const Weights w = a.w;
  constexpr int kStride = (MODE == 0) ? kCoefS : kCoefT;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kStride;
  const float* __restrict__ x0 = a.x0 + p * 3;
  const int c = static_cast<int>(local);
  float* su = s_rec + 4;
  float* sx = s_rec + 4 + 2 * n;
  // where the winner's controls come from: the control matrix, or - regenerate - its global index alone
  const uint32_t gidx = static_cast<uint32_t>(key & 0xffffffffLL);
  float z[kKnots][2] = {};
  float amp = 0.0f;
  const float* centre = nullptr;
  if (a.regenerate) {
    draw_normals(a.spec, gidx, static_cast<uint32_t>(p), z);
    const float* alt = (gidx == 1u) ? a.u_ref : (gidx == 2u) ? a.u_extra : nullptr;
    const bool use_ref = alt != nullptr;
    amp = use_ref ? 0.0f : candidate_amplitude(gidx);
    centre = use_ref ? alt + static_cast<size_t>(p) * n * 2 : a.centre + static_cast<size_t>(p) * a.centre_stride;
  }
  auto winner_control = [&](int step, float& v, float& k) {
    if (a.regenerate) {
      regenerate_control(a.spec, z, amp, centre, step, v, k);
    } else {
      float vv[1], kk[1];
      load_controls<LAYOUT, 1>(a.U, p, a.N, n, step, c, vv, kk);
      v = vv[0];
      k = kk[0];
    }
  };

  if (a.controls_only) {
    // between the rounds of an optimisation only the winner's controls are needed (the next round's centre):
    // header = [cost from the key, 0, n_feasible, 1], u block, no re-roll
    for (int step = lane; step < n; step += kWave) {
      float v, k;
      winner_control(step, v, k);
      su[2 * step] = v;
      su[2 * step + 1] = k;
    }
    if (lane == 0) {
      s_rec[0] = key_cost(key);
      s_rec[1] = 0.0f;
      s_rec[2] = static_cast<float>(nfeas);
      s_rec[3] = 1.0f;
    }
    __syncthreads();
    for (int e = lane; e < 4 + 2 * n; e += kWave) rec[e] = s_rec[e];
    return;
  }

  if constexpr (MODE == 0) {
    // Re-roll split by dependence: only the 9-operation state recurrence is sequential (run by the whole wave in
    // lock-step on broadcast inputs); everything that hangs off a state - stage cost, the four bound terms - is
    // evaluated by lane i for step i in parallel, and the partial results are then accumulated in step order so
    // that J and V see exactly the additions of step_spatial(), in the same order.
    float ey = x0_first[0], ep = x0_first[1], t = x0_first[2], J = 0.0f, V = 0.0f;
    for (int base = 0; base < n; base += kWave) {
      const int mine = base + lane;
      const bool valid = mine < n;
      float v[1] = {0.0f}, k[1] = {0.0f};
      float row[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (valid) {
        winner_control(mine, v[0], k[0]);
#pragma unroll
        for (int q = 0; q < 9; ++q) row[q] = (base == 0) ? row_first[q] : coef[mine * kCoefS + q];
        su[2 * mine] = v[0];
        su[2 * mine + 1] = k[0];
      }
      const float dv = v[0] - row[5];
      const float dk = k[0] - row[6];
      const float term_ep = row[0] * dk;   // ds * dk
      const float term_t = row[3] * dv;    // b31 * dv
      float r = quad(w.r0, dv);
      r = r + quad(w.r1, dk);
      const float hu0 = hinge2(w.ulo0 - v[0], v[0] - w.uhi0);
      const float hu1 = hinge2(w.ulo1 - k[0], k[0] - w.uhi1);
      float my_ey = 0.0f, my_ep = 0.0f, my_t = 0.0f, nx_ey = 0.0f, nx_t = 0.0f;
      const int steps = min(kWave, n - base);
      for (int i = 0; i < steps; ++i) {
        const float b_ds = bcast(row[0], i), b_a21 = bcast(row[1], i), b_a31 = bcast(row[2], i);
        const float b_f3 = bcast(row[4], i), b_te = bcast(term_ep, i), b_tt = bcast(term_t, i);
        const float ey_n = ey + b_ds * ep;
        const float ep_n = (ep + b_a21 * ey) + b_te;
        const float t_n = ((t + b_a31 * ey) + b_tt) + b_f3;
        if (lane == i) {
          my_ey = ey;
          my_ep = ep;
          my_t = t;
          nx_ey = ey_n;
          nx_t = t_n;
        }
        ey = ey_n;
        ep = ep_n;
        t = t_n;
      }
      float q = quad(w.q0, my_ey);
      q = q + quad(w.q1, my_ep);
      q = q + quad(w.q2, my_t);
      const float stage = 0.5f * (q + r);
      const float hc = hinge2(row[7] - nx_ey, nx_ey - row[8]);
      const float tv = fmaxf(w.tmin - nx_t, 0.0f);
      const float ht = tv * tv;
      if (valid) {
        sx[3 * mine] = my_ey;
        sx[3 * mine + 1] = my_ep;
        sx[3 * mine + 2] = my_t;
      }
      for (int i = 0; i < steps; ++i) {
        J = J + bcast(stage, i);
        V = V + bcast(hu0, i);
        V = V + bcast(hu1, i);
        V = V + bcast(hc, i);
        V = V + bcast(ht, i);
      }
    }
    if (lane == 0) {
      sx[3 * n] = ey;
      sx[3 * n + 1] = ep;
      sx[3 * n + 2] = t;
      const StateS st{ey, ep, t, J, V};
      s_rec[0] = finish_spatial(st, w);
      s_rec[1] = V;
    }
  } else {
    StateT st = start_temporal<float>(x0, coef);
    // waypoint table -> LDS (behind the record image); each lane also keeps "its" waypoint's (x, y) in registers
    float* s_wp = s_rec + ((rec_floats + 3) & ~3);
    float* s_abc = s_wp + n * kCoefT;
    stage_temporal_tables(coef, n, lane, kWave, s_wp, s_abc);
    if (lane == 0) {
      sx[0] = st.X + coef[0];   // (poses leave in the caller's frame: start_temporal())
      sx[1] = st.Y + coef[1];
      sx[2] = st.phi;
    }
    __syncthreads();
    const int mine_at = kKeyStride * min(lane, n - 1);   // the search key's entries of "this lane's" waypoint
    const float my_a = s_abc[mine_at], my_b = s_abc[mine_at + kKeyB], my_c = s_abc[mine_at + kKeyC];
    int j_prev = 0;
    for (int base = 0; base < n; base += kWave) {
      const int mine = base + lane;
      float v[1] = {0.0f}, k[1] = {0.0f};
      if (mine < n) {
        winner_control(mine, v[0], k[0]);
        su[2 * mine] = v[0];
        su[2 * mine + 1] = k[0];
      }
      const int steps = min(kWave, n - base);
      for (int i = 0; i < steps; ++i) {
        const float vi = bcast(v[0], i), ki = bcast(k[0], i);
        temporal_advance(st, vi, ki, w);
        // nearest waypoint with the lanes scanning the table side by side; (distance, index) keys keep the
        // first minimum exactly like the one-lane scan of the rollout kernel
        // (a search window, when configured, only masks out the waypoints outside it)
        const int win_w = w.nn_back + w.nn_ahead + 1;
        const int win_lo = (w.nn_ahead < 0) ? 0 : max(min(j_prev - w.nn_back, n - win_w), 0);
        const int win_hi = (w.nn_ahead < 0) ? n - 1 : min(win_lo + win_w, n) - 1;
        float best = (lane >= win_lo && lane <= win_hi) ? search_key<float>(st.X, st.Y, my_a, my_b, my_c) : __builtin_inff();
        int j = lane;
        for (int m = lane + kWave; m <= win_hi; m += kWave) {
          const float d = (m >= win_lo) ? search_key<float>(st.X, st.Y, s_abc[kKeyStride * m], s_abc[kKeyStride * m + kKeyB], s_abc[kKeyStride * m + kKeyC])
                                        : __builtin_inff();
          const bool better = d < best;
          best = better ? d : best;
          j = better ? m : j;
        }
#pragma unroll
        for (int mask = 32; mask >= 1; mask >>= 1) {
          const float ob = __shfl_xor(best, mask, kWave);
          const int oj = __shfl_xor(j, mask, kWave);
          const bool take = (ob < best) || (ob == best && oj < j);
          best = take ? ob : best;
          j = take ? oj : j;
        }
        j = __builtin_amdgcn_readfirstlane(j);
        j_prev = j;
        temporal_cost(st, s_wp + j * kCoefT, vi, ki, w);
        if (lane == 0) {
          sx[3 * (base + i + 1)] = st.X + coef[0];
          sx[3 * (base + i + 1) + 1] = st.Y + coef[1];
          sx[3 * (base + i + 1) + 2] = st.phi;
        }
      }
    }
    if (lane == 0) {
      s_rec[0] = finish_temporal(st, n, w);
      s_rec[1] = st.V;
    }
  }
  if (lane == 0) {
    s_rec[2] = static_cast<float>(nfeas);
    s_rec[3] = 1.0f;
  }
  __syncthreads();
  for (int e = lane; e < rec_floats; e += kWave) rec[e] = s_rec[e];
}

template <int MODE, int LAYOUT>
__global__ void __launch_bounds__(kWave) finalize_kernel(const FinalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_finalize[];
  finalize_problem<MODE, LAYOUT>(a, blockIdx.x, s_finalize);
}

constexpr int kGroupFinalizeProblems = 256;   // from this many problems the batched finalize runs sixteen lanes per problem

// The batched solve's finalize, mode S, from 256 problems up (round 4): SIXTEEN LANES PER PROBLEM, four problems per
// wavefront.  finalize_kernel gives every problem a wavefront, which re-rolls its winner in lock-step on broadcast inputs - a
// fine use of a wave when there is one problem, and 4 096 lone-wave walks when there are 4 096: 29 us behind the headline's
// 1.0 ms rollout.  (A LANE per problem - the argmin, the winner's controls and table rows fetched seven steps ahead,
// step_spatial() per lane - was built first: 1/64 of the instructions, but every load and store of a wave goes to 64
// different cache lines and step_spatial's ~45 instructions per step run on 64 lone waves: 20.5 us.)
// finalize_problem()'s decomposition does better on a quarter of a wave per problem:
//   A  lane l of the problem's sixteen prepares steps l, l + 16, ...: the winner's controls (re-drawn - the four Philox
//      blocks of its normals drawn by four of the lanes and shared through LDS - or loaded), the table row, everything of
//      the step that does not depend on the state; the record's u block
//   B  the 9-operation state recurrence, the only sequential part, on operands read from LDS eight steps ahead
//   C  lane l again: stage cost and the two state-dependent bound terms of its steps from the states B left in LDS
//   D  J and V accumulated in step order: the additions of step_spatial(), in its order - the same bits
// and the four record images, contiguous in LDS as the four records are in memory, leave in 16-byte stores: 13.8 us.
// Fewer problems leave the CUs idle either way and keep the wave-per-problem form, whose latency is the shorter.
constexpr int kGroupLanes = 16;
constexpr int kGroupProblems = kWave / kGroupLanes;
constexpr int kGroupRow = 8;     // floats per step in each of the two LDS tables
constexpr int kGroupAhead = 8;   // steps whose LDS operands are requested together in B and D (the kernel of its own)
constexpr int kGroupPass = 4;    // steps per lane whose memory operands are requested together in A (the kernel of its own)

__device__ __forceinline__ int group_finalize_floats_device(int n) {   // one problem per wavefront
  return ((4 + 2 * n + 3 * (n + 1) + 3) & ~3) + 16 + 2 * n * kGroupRow;
}
inline size_t group_finalize_floats(int n, int problems = kGroupProblems) {   // LDS floats of ONE wavefront
  const size_t records = (static_cast<size_t>(problems) * (4 + 2 * n + 3 * (n + 1)) + 3) & ~static_cast<size_t>(3);
  return records + problems * (16 + 2 * static_cast<size_t>(n) * kGroupRow);
}

// finalize_group runs on ONE wavefront: what its lanes hand each other through LDS needs no workgroup barrier (a wave's LDS
// instructions execute in order), only the compiler kept from moving the accesses across the hand-off - so the function
// is the same inside a 256-thread workgroup whose other waves have retired (rollout_chained_kernel)
__device__ __forceinline__ void wave_lds_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wavefront's four problems: threadIdx.x < 64 are its lanes, `group` its place in the batch.  kPass / kAhead: how far
// ahead operands are requested (registers: 116 with 4 / 8, under 64 with 1 / 4 - what a launch shared with the rollout,
// rollout_chained_kernel, can afford).
template <int LAYOUT, int kPass, int kAhead, int kLanes = kGroupLanes>
__device__ __forceinline__ void finalize_group(const FinalizeArgs& a, const int group, float* s_group) {
  constexpr int kProblems = kWave / kLanes;   // problems of this wavefront
  const int lane = static_cast<int>(threadIdx.x) & (kWave - 1);
  const int q = lane / kLanes, sub = lane % kLanes;
  const int n = a.n;
  const int rec_floats = 4 + 2 * n + 3 * (n + 1);
  const int p_first = group * kProblems;
  const int p = p_first + q;
  const bool live = p < a.P;
  const int pl = live ? p : a.P - 1;   // a quarter without a problem repeats the last one's work and writes nothing
  const float* __restrict__ x0 = a.x0 + pl * 3;
  float ey = 0.0f, ep = 0.0f, t = 0.0f;
  float rows_ahead[kPass][9] = {}, seg_ahead[kPass][2] = {}, cen_ahead[kPass][2] = {};
  if (a.records != nullptr) {   // what does not depend on who won travels with the partial keys
    ey = x0[0], ep = x0[1], t = x0[2];
    const float* __restrict__ coef_ahead = a.coef + static_cast<size_t>(pl) * n * kCoefS;
#pragma unroll
    for (int j = 0; j < kPass; ++j) {
      const int i = min(j * kLanes + sub, n - 1);
#pragma unroll
      for (int e = 0; e < 9; ++e) rows_ahead[j][e] = coef_ahead[i * kCoefS + e];
      if (a.regenerate) {   // (the centre every winner but candidates 1 and 2 is drawn round: those two fetch theirs below)
        seg_ahead[j][0] = a.spec.segments[2 * i];
        seg_ahead[j][1] = a.spec.segments[2 * i + 1];
        cen_ahead[j][0] = a.centre[static_cast<size_t>(pl) * a.centre_stride + 2 * i];
        cen_ahead[j][1] = a.centre[static_cast<size_t>(pl) * a.centre_stride + 2 * i + 1];
      }
    }
  }

  int nfeas = 0;
  int64_t key = kKeyMax;
  for (int b = sub; b < a.blocks_per_problem; b += kLanes) {
    const size_t slot = static_cast<size_t>(pl) * a.blocks_per_problem + b;
    nfeas += a.partial_feas[slot];
    const int64_t kb = a.partial_keys[slot];
    key = (kb < key) ? kb : key;
  }
#pragma unroll
  for (int m = kLanes / 2; m >= 1; m >>= 1) {
    nfeas += __shfl_xor(nfeas, m, kLanes);
    const int64_t other = static_cast<int64_t>(__shfl_xor(static_cast<long long>(key), m, kLanes));
    key = (other < key) ? other : key;
  }
  if (a.keys_in != nullptr) key = a.keys_in[pl];
  if (a.keys_out != nullptr && live && sub == 0) a.keys_out[p] = key;
  if (a.records == nullptr) return;

  float* s_rec = s_group + q * rec_floats;
  float* s_tables = s_group + ((kProblems * rec_floats + 3) & ~3);
  float* s_z = s_tables + q * 16;
  float* s_rows = s_tables + kProblems * 16 + q * n * kGroupRow;                      // ds a21 a31 f3 | ds dk, b31 dv, lo, hi
  float* s_terms = s_tables + kProblems * 16 + (kProblems + q) * n * kGroupRow;   // stage (r), hu0, hu1, hc | ht
  float* su = s_rec + 4;
  float* sx = s_rec + 4 + 2 * n;
  const int64_t local = static_cast<int64_t>(static_cast<uint32_t>(key & 0xffffffffLL)) - a.index_offset;
  const bool owner = a.regenerate || (local >= 0 && local < a.N);
  const int c = owner ? static_cast<int>(local) : 0;
  const uint32_t gidx = static_cast<uint32_t>(key & 0xffffffffLL);
  const Weights w = a.w;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(pl) * n * kCoefS;
  float amp = 0.0f;
  const float* centre = nullptr;
  bool use_alt = false;
  if (a.regenerate) {
    if (sub < kKnots / 2) {
      float four[4];
      draw_normal_block(a.spec, gidx, static_cast<uint32_t>(pl), static_cast<uint32_t>(sub), four);
      *reinterpret_cast<f32x4*>(s_z + 4 * sub) = f32x4{four[0], four[1], four[2], four[3]};
    }
    const float* alt = (gidx == 1u) ? a.u_ref : (gidx == 2u) ? a.u_extra : nullptr;
    use_alt = alt != nullptr;
    amp = use_alt ? 0.0f : candidate_amplitude(gidx);
    centre = use_alt ? alt + static_cast<size_t>(pl) * n * 2 : a.centre + static_cast<size_t>(pl) * a.centre_stride;
  }
  wave_lds_handoff();

  // A: everything of a step that does not depend on the state, four steps per lane and pass; a pass's operands are all
  // requested before the first is used (the first pass's table rows before the keys: rows_ahead above)
  for (int base = 0; base < n; base += kPass * kLanes) {
    float rows[kPass][9], v[kPass], k[kPass], seg[kPass][2], cen[kPass][2];
#pragma unroll
    for (int j = 0; j < kPass; ++j) {
      const int i = min(base + j * kLanes + sub, n - 1);
      if (a.regenerate) {
        if (base == 0) {
          seg[j][0] = seg_ahead[j][0];
          seg[j][1] = seg_ahead[j][1];
        } else {
          seg[j][0] = a.spec.segments[2 * i];
          seg[j][1] = a.spec.segments[2 * i + 1];
        }
        if (base == 0 && !use_alt) {   // (per lane: a quarter whose winner is candidate 1 or 2 fetches its own centre)
          cen[j][0] = cen_ahead[j][0];
          cen[j][1] = cen_ahead[j][1];
        } else {
          cen[j][0] = centre[2 * i];
          cen[j][1] = centre[2 * i + 1];
        }
      } else {
        float vv[1], kk[1];
        load_controls<LAYOUT, 1>(a.U, pl, a.N, n, i, c, vv, kk);
        v[j] = vv[0];
        k[j] = kk[0];
      }
#pragma unroll
      for (int e = 0; e < 9; ++e) rows[j][e] = (base == 0) ? rows_ahead[j][e] : coef[i * kCoefS + e];
    }
#pragma unroll
    for (int j = 0; j < kPass; ++j) {
      const int i = base + j * kLanes + sub;
      if (i < n) {
        if (a.regenerate) {   // regenerate_control()'s operands, the bracketing knots read from LDS
          const int k0 = static_cast<int>(seg[j][0]);
          const int knot = (k0 >= 1 && k0 <= kKnots - 2) ? k0 : 0;
          const float* z = s_z + 2 * knot;
          blend_control(a.spec, amp, seg[j][1], cen[j][0], cen[j][1], z[0], z[1], z[2], z[3], v[j], k[j]);
        }
        const float* row = rows[j];
        const float dv = v[j] - row[5];
        const float dk = k[j] - row[6];
        const float term_ep = row[0] * dk;
        const float term_t = row[3] * dv;
        float r = quad(w.r0, dv);
        r = r + quad(w.r1, dk);
        const float hu0 = hinge2(w.ulo0 - v[j], v[j] - w.uhi0);
        const float hu1 = hinge2(w.ulo1 - k[j], k[j] - w.uhi1);
        *reinterpret_cast<f32x4*>(s_rows + i * kGroupRow) = f32x4{row[0], row[1], row[2], row[4]};
        *reinterpret_cast<f32x4*>(s_rows + i * kGroupRow + 4) = f32x4{term_ep, term_t, row[7], row[8]};
        *reinterpret_cast<f32x4*>(s_terms + i * kGroupRow) = f32x4{r, hu0, hu1, 0.0f};
        su[2 * i] = v[j];
        su[2 * i + 1] = k[j];
      }
    }
  }
  wave_lds_handoff();

  // B: the recurrence (every lane of the quarter runs it on the same LDS words; lane 0 of the quarter leaves the states)
  for (int base = 0; base < n; base += kAhead) {
    f32x4 lo[kAhead];
    float te[kAhead], tt[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int i = min(base + j, n - 1);
      lo[j] = *reinterpret_cast<const f32x4*>(s_rows + i * kGroupRow);
      te[j] = s_rows[i * kGroupRow + 4];
      tt[j] = s_rows[i * kGroupRow + 5];
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int i = base + j;
      if (i < n) {
        if (sub == 0) {
          sx[3 * i] = ey;
          sx[3 * i + 1] = ep;
          sx[3 * i + 2] = t;
        }
        const float ey_n = ey + lo[j][0] * ep;
        const float ep_n = (ep + lo[j][1] * ey) + te[j];
        const float t_n = ((t + lo[j][2] * ey) + tt[j]) + lo[j][3];
        ey = ey_n;
        ep = ep_n;
        t = t_n;
      }
    }
  }
  if (sub == 0) {
    sx[3 * n] = ey;
    sx[3 * n + 1] = ep;
    sx[3 * n + 2] = t;
  }
  wave_lds_handoff();

  // C: what hangs off a state
  for (int base = 0; base < n; base += kLanes) {
    const int i = base + sub;
    if (i < n) {
      const float my_ey = sx[3 * i], my_ep = sx[3 * i + 1], my_t = sx[3 * i + 2];
      const float nx_ey = sx[3 * i + 3], nx_t = sx[3 * i + 5];
      float s = quad(w.q0, my_ey);
      s = s + quad(w.q1, my_ep);
      s = s + quad(w.q2, my_t);
      const float stage = 0.5f * (s + s_terms[i * kGroupRow]);
      const float hc = hinge2(s_rows[i * kGroupRow + 6] - nx_ey, nx_ey - s_rows[i * kGroupRow + 7]);
      const float tv = fmaxf(w.tmin - nx_t, 0.0f);
      s_terms[i * kGroupRow] = stage;
      s_terms[i * kGroupRow + 3] = hc;
      s_terms[i * kGroupRow + 4] = tv * tv;
    }
  }
  wave_lds_handoff();

  // D: step_spatial()'s additions in its order
  float J = 0.0f, V = 0.0f;
  for (int base = 0; base < n; base += kAhead) {
    f32x4 first[kAhead];
    float ht[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int i = min(base + j, n - 1);
      first[j] = *reinterpret_cast<const f32x4*>(s_terms + i * kGroupRow);
      ht[j] = s_terms[i * kGroupRow + 4];
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      if (base + j < n) {
        J = J + first[j][0];
        V = V + first[j][1];
        V = V + first[j][2];
        V = V + first[j][3];
        V = V + ht[j];
      }
    }
  }
  wave_lds_handoff();   // (the terms are read: a problem whose winner lives on another rank may now blank its image)
  if (owner) {
    if (sub == 0) {
      const StateS st{ey, ep, t, J, V};
      s_rec[0] = finish_spatial(st, w);
      s_rec[1] = V;
      s_rec[2] = static_cast<float>(nfeas);
      s_rec[3] = 1.0f;
    }
  } else {
    for (int e = sub; e < rec_floats; e += kLanes) s_rec[e] = (e == 2) ? static_cast<float>(nfeas) : 0.0f;
  }
  wave_lds_handoff();
  const int count = min(kProblems, a.P - p_first) * rec_floats;
  float* __restrict__ out = a.records + static_cast<size_t>(p_first) * rec_floats;
  const int whole = ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? (count & ~3) : 0;
  for (int e = 4 * lane; e < whole; e += 4 * kWave) *reinterpret_cast<f32x4*>(out + e) = *reinterpret_cast<const f32x4*>(s_group + e);
  for (int e = whole + lane; e < count; e += kWave) out[e] = s_group[e];
}

template <int LAYOUT>
__global__ void __launch_bounds__(kWave) finalize_groups_kernel(const FinalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_group[];
  finalize_group<LAYOUT, kGroupPass, kGroupAhead>(a, static_cast<int>(blockIdx.x), s_group);
}

// The same with a whole wavefront per problem and four wavefronts per workgroup (A/B: ACMPC_FINALIZE_WAVES=1; and the
// form the finalize takes inside rollout_chained_kernel): a lane per step in A and C - one pass up to 64 steps, every
// operand requested before the keys are reduced -, B and D as before.
constexpr int kFinalizeWaves = 4;

template <int LAYOUT>
__global__ void __launch_bounds__(kFinalizeWaves * kWave) finalize_waves_kernel(const FinalizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_group[];
  const int wave = static_cast<int>(threadIdx.x) / kWave;
  const int p = static_cast<int>(blockIdx.x) * kFinalizeWaves + wave;
  if (p >= a.P) return;
  finalize_group<LAYOUT, 1, kGroupAhead, kWave>(a, p, s_group + wave * group_finalize_floats_device(a.n));
}

// A STREAM of batches (acmpc_solve_stream_device): batch k's rollout and batch k - 1's finalize in ONE launch.  Behind
// the headline's rollout the finalize of its 4 096 problems is 13-14 us of lone waves plus a launch boundary - 1.7 % of the
// step - and no form of it is fast enough to vanish (a lone wave issues an instruction every 4.2 cycles: finalize_group's
// ~2 400 are 5 us before the first cache miss).  So it runs where nobody waits for it: the LAST rows of the next rollout's
// grid are rows of finalize workgroups - four wavefronts, a problem each, a lane per step (finalize_group<..., 64>: one
// pass, every operand requested before the keys are reduced, two trips to memory).  Rows are dispatched in order, so these
// start when the grid has no rollout workgroup left to hand out, in the wave slots the rollout's last generation leaves
// empty as it drains, and are done before it is.  Measured on the headline's batch (tools/archive/chained_ab.py, launches
// alternating with the plain kernel in one process; tools/archive/finalize_ab.sh, interleaved bench runs): the kernel is as long
// as the plain one or up to 10 us longer (by the box: how ragged the rollout's tail is), the step 2-15 us shorter than
// with two launches in every pair.  What did NOT work, same tools: the
// sixteen-lane form on ONE wave per finalize workgroup in the same last rows, +11.5 us on the kernel (its six dependent
// trips to memory outlast the tail: the step as long as with two launches); the same rows spread evenly through the grid
// (one in seventeen), +25 us - every finalize workgroup then displaces a rollout workgroup for its ~40 us under a saturated
// memory system, 1 024 of them 19 us of the 2 048 resident workgroups' time; raising the lone waves' priority (s_setprio)
// changed nothing, looking further ahead for operands (more registers: spills under the rollout's 64) made it worse.  The
// two batches' partial keys live in different halves of the handle's buffer.
#ifndef ACMPC_CHAINED_PASS
#define ACMPC_CHAINED_PASS 1    // (A/B builds) finalize_group's look-ahead inside the shared launch
#endif
#ifndef ACMPC_CHAINED_AHEAD
#define ACMPC_CHAINED_AHEAD 4
#endif
template <int LAYOUT, int CPT, int BLOCK, int PACK>
__global__ void __launch_bounds__(BLOCK, 8) rollout_chained_kernel(const RolloutArgs a, const FinalizeArgs f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int y = static_cast<int>(blockIdx.y);
  if (y >= a.P) {
    static_assert(BLOCK == kFinalizeWaves * kWave, "a finalize workgroup is four wavefronts, a problem each");
    const int wave = static_cast<int>(threadIdx.x) / kWave;
    const int p = ((y - a.P) * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x)) * kFinalizeWaves + wave;
    if (p >= f.P) return;
    finalize_group<1, ACMPC_CHAINED_PASS, ACMPC_CHAINED_AHEAD, kWave>(
        f, p, reinterpret_cast<float*>(smem) + wave * group_finalize_floats_device(f.n));
    return;
  }
  rollout_block<0, LAYOUT, CPT, BLOCK, PACK, false>(a, smem, y);
}

// Record of problem p out of the winning workgroup's trace (see rollout_sampled_body): argmin over the partial keys,
// then a copy - no arithmetic, so the record holds exactly the bits the winning lane computed, which are the bits a
// re-roll of that candidate computes.  One wave.
__device__ __forceinline__ void finalize_from_trace(const RolloutArgs& a, const FusedFinalize& fused, const int p) {
  const int lane = threadIdx.x;
  const int n = a.n;
  const int blocks = static_cast<int>(gridDim.x);
  // every key is requested before the first is looked at (a load per loop iteration with its compare behind it is one
  // round trip per 64 workgroups: 3 us of a 256-workgroup round's tail); launches beyond the batch go round the loop
  constexpr int kBatch = 4;
  int nfeas = 0;
  int64_t key = kKeyMax;
  for (int b0 = 0; b0 < blocks; b0 += kBatch * kWave) {
    int64_t kb[kBatch];
    int fb[kBatch];
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      kb[q] = kKeyMax;
      fb[q] = 0;
      if (b0 + q * kWave < blocks) {   // (wave-uniform)
        const size_t slot = static_cast<size_t>(p) * blocks + min(b0 + q * kWave + lane, blocks - 1);
        kb[q] = observe(&a.partial_keys[slot]);
        fb[q] = observe(&a.partial_feas[slot]);
      }
    }
#pragma unroll
    for (int q = 0; q < kBatch; ++q) {
      const bool mine = b0 + q * kWave + lane < blocks;   // (a clamped lane re-read the last workgroup's slot)
      key = (mine && kb[q] < key) ? kb[q] : key;
      nfeas += mine ? fb[q] : 0;
    }
  }
  nfeas = wave_sum_int(nfeas);
  const int64_t best = wave_min_key(key);
  // workgroup b holds the candidates b * 64 ..: the winner's workgroup follows from its index
  const int block = static_cast<int>(static_cast<int64_t>(static_cast<uint32_t>(best & 0xffffffffLL)) - a.index_offset) / kWave;
  const float* trace = fused.trace + (static_cast<size_t>(p) * blocks + block) * fused.trace_pitch;
  const int rec_floats = 4 + 2 * n + 3 * (n + 1);
  float* __restrict__ rec = fused.records + static_cast<size_t>(p) * rec_floats;
  const float* __restrict__ x0 = a.x0 + p * 3;
  const int count = fused.controls_only ? 4 + 2 * n : rec_floats;
  // every entry but three is ONE load from an address that depends on the entry alone: all of a pass requested, then stored
  constexpr int kSlots = 8;
  for (int e0 = 0; e0 < count; e0 += kSlots * kWave) {
    float value[kSlots];
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      if (e0 + q * kWave < count) {   // (wave-uniform)
        const int e = min(e0 + q * kWave + lane, count - 1);
        const float* src = trace + (e - 7);                               // states: 2n + (e - 4 - 2n - 3)
        src = (e < 4 + 2 * n + 3) ? x0 + (e - (4 + 2 * n)) : src;         // start state
        src = (e < 4 + 2 * n) ? trace + (e - 4) : src;                    // controls
        src = (e < 4) ? trace + 5 * n + (1 - min(e, 1)) : src;            // cost, violation
        value[q] = observe(src);
      }
    }
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      const int e = e0 + q * kWave + lane;
      if (e0 + q * kWave < count) {
        float out = value[q];
        // between rounds the cost is read back from the key (non-finite -> +inf), as the separate finalize does; the
        // last record carries the lane's own
        if (fused.controls_only) out = (e == 0) ? key_cost(best) : (e == 1) ? 0.0f : out;
        out = (e == 2) ? static_cast<float>(nfeas) : (e == 3) ? 1.0f : out;
        if (e < count) rec[e] = out;
      }
    }
  }
}

// Last-workgroup-done, in two levels: a workgroup publishes its partials (and trace) device-wide and takes a ticket of
// its group (workgroup index mod `groups`); the last of a group takes a ticket of the problem; the last of those knows
// every workgroup's results are at the coherence point (each waited for its stores before its increment) and
// finalizes.  Two levels because a device-scope atomic on one address takes ~13 ns and they serialise: 256 workgroups
// finishing together would queue for 3 us on one counter, and queue for 0.5 us on 8 + 1.  `tickets` - this problem's
// [groups + 1] counters, kTicketStride ints apart - is zero before the launch and after it.  Called by one whole wave whose threadIdx.x are its
// lanes; true (wave-uniform) on the wave that may read what the others published.
//
// Memory model: the values that cross workgroups are written and read with relaxed agent-scope atomics and ordered by
// s_waitcnt vmcnt(0) on the writer's side (published()) and by the ticket's data dependence on the reader's.  That
// relies on gfx942 / gfx950 hardware - an agent-scope (sc1) store is acknowledged only once it is at the memory-side
// coherence point, and agent-scope loads are served from there - not on the HSA memory model, under which it is a
// data race.  The signal fences keep the COMPILER from moving the reader's loads above the ticket.  On any other
// architecture: ACMPC_NO_CHAINED_ROUNDS / ACMPC_NO_TRACED_FINALIZE / ACMPC_NO_SOLO select the forms without it.
__device__ __forceinline__ bool last_workgroup_of_problem(int* tickets, const int groups_cfg) {
  const int blocks = static_cast<int>(gridDim.x);
  const int group = static_cast<int>(blockIdx.x) & (groups_cfg - 1);   // groups_cfg is a power of two
  const int group_size = (blocks - group + groups_cfg - 1) / groups_cfg;
  const int groups = min(blocks, groups_cfg);
  published();   // partial key, feasible count and trace of this workgroup are at the coherence point
  ACMPC_STAMP(5);
  int ticket = 0;
  if (threadIdx.x == 0) ticket = atomicAdd(&tickets[group * kTicketStride], 1);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (ticket != group_size - 1) return false;
  ACMPC_STAMP(6);
  if (threadIdx.x == 0) {
    publish(&tickets[group * kTicketStride], 0);
    ticket = atomicAdd(&tickets[groups_cfg * kTicketStride], 1);
  }
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (ticket != groups - 1) return false;
  ACMPC_STAMP(7);
  if (threadIdx.x == 0) publish(&tickets[groups_cfg * kTicketStride], 0);  // the launch leaves the counters as it found them
  return true;
}

// rollout_kernel with the finalize in its own launch (round 4): the workgroup that finishes a problem LAST (tickets, as in
// the fused rounds) also takes the argmin over the problem's partial keys and writes the winner's record - re-rolled by
// its first wave with the step functions of the rollout, as finalize_kernel does (finalize_problem: the same bits).  The
// batched solve (4 096 problems x 4 096 candidates) used to follow its 1.09 ms rollout with a finalize_kernel of 4 096
// single-wave workgroups: 29 us + a launch gap, 3.4 % of the step.  Here the re-rolls of all problems but the last few
// run in the shadow of other workgroups' streaming (the kernel is HBM-bound: the vector pipes have room), and only the
// final ones - ~9 us of a lone wave - are exposed.  Mode S, step-major, 256-thread workgroups; one candidate's states per
// lane would not fit the headline kernel's 46 registers, so no trace: the winner is rolled again.
// LDS: [64 bytes: wave keys, counts] [record image 4 + 2n + 3(n + 1) floats].
#ifndef ACMPC_TAILED_WAVES
#define ACMPC_TAILED_WAVES 8   // waves per SIMD asked for: caps the allocation at 64 VGPRs (the tail's re-roll would take 78)
#endif
template <int MODE, int LAYOUT, int CPT, int BLOCK, int PACK>
__global__ void __launch_bounds__(BLOCK, ACMPC_TAILED_WAVES) rollout_tailed_kernel(const RolloutArgs a, const FinalizeArgs f, int* tickets) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  rollout_block<MODE, LAYOUT, CPT, BLOCK, PACK, true>(a, smem, static_cast<int>(blockIdx.y));
  if (threadIdx.x >= kWave) return;   // the tail is the first wave's (the one that published the workgroup's partials)
  const int p = blockIdx.y;
  if (!last_workgroup_of_problem(tickets + static_cast<size_t>(p) * (kTicketGroups + 1) * kTicketStride, kTicketGroups)) return;
  finalize_problem<MODE, LAYOUT>(f, p, reinterpret_cast<float*>(smem + 64));
}

// Optional tail of a fused round (one wave, threadIdx.x = its lanes): see FusedFinalize.
template <int MODE>
__device__ __forceinline__ void fused_tail(const RolloutArgs& a, const SampleArgs& smp, const FusedFinalize& fused,
                                           const bool traced, float* s_finalize) {
  const int p = blockIdx.y;
  if (fused.tickets == nullptr) return;
  if (!last_workgroup_of_problem(fused.tickets + static_cast<size_t>(p) * (kTicketGroups + 1) * kTicketStride, kTicketGroups)) return;
  auto signal_done = [&]() {   // the record (written by all lanes) before the flag (lane 0), both on their way to the host
    if (fused.done == nullptr || p != 0) return;
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(fused.done, fused.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (traced) {
    finalize_from_trace(a, fused, p);
    signal_done();
    return;
  }
  FinalizeArgs f{};
  f.x0 = a.x0;
  f.coef = a.coef;
  f.partial_keys = a.partial_keys;
  f.partial_feas = a.partial_feas;
  f.records = fused.records;
  f.regenerate = true;
  f.controls_only = fused.controls_only;
  f.centre = smp.centre;
  f.u_ref = smp.u_ref;
  f.u_extra = smp.u_extra;
  f.centre_stride = smp.centre_stride;
  f.spec = smp.spec;
  f.blocks_per_problem = static_cast<int>(gridDim.x);
  f.P = a.P;
  f.N = a.N;
  f.n = a.n;
  f.index_offset = a.index_offset;
  f.w = a.w;
  finalize_problem<MODE, 1>(f, p, s_finalize);
  signal_done();
}

// LDS: [rollout part: mode T tables] [trace [5n + 2][64], or - no trace - the finalize's record image (+ mode T table)]
//      [uniform operands of the steps: sampled_uniform_floats()]
template <int MODE>
__global__ void __launch_bounds__(kWave) rollout_sampled_kernel(const RolloutArgs a, const SampleArgs smp,
                                                                const FusedFinalize fused, const int rollout_lds_floats,
                                                                const int uniform_lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float s_fused[];
  const int p = blockIdx.y;
  const bool traced = fused.trace != nullptr;
  rollout_sampled_body<MODE>(a, smp, s_fused, uniform_lds_floats, traced, rollout_lds_floats,
                             fused.trace + (static_cast<size_t>(p) * gridDim.x + blockIdx.x) * fused.trace_pitch);
  fused_tail<MODE>(a, smp, fused, traced, s_fused + rollout_lds_floats);
}

// The traced mode-S round on TWO waves per workgroup.  A round at the closed-loop size is one wave per CU walking a serial
// stream of ~5 700 instructions, three SIMDs of its CU idle, and the stream falls into two parts that only meet in
// (v, kappa): drawing the normals and blending them into the controls of each step, and rolling those controls.  Here
// wave 1 - the producer - draws and blends, running ahead through the horizon and handing (v, kappa) per step and lane
// over through LDS in chunks of kPairChunk steps, two buffers; wave 0 - the consumer - rolls them (step_spatial, the
// same call) one chunk behind, one barrier per chunk, and then reduces, publishes and runs the tail as the single-wave
// kernel does.  Same operations on the same operands: same bits (test_optimize_forms_agree).
// LDS: [trace [5n + 2][64]] [uniform operands] [exchange [2][kPairChunk][2][64]].
constexpr int kPairChunk = 7;
constexpr int kPairValues = 2;

__global__ void __launch_bounds__(2 * kWave) rollout_sampled_pair_kernel(const RolloutArgs a, const SampleArgs smp,
                                                                         const FusedFinalize fused,
                                                                         const int uniform_lds_floats,
                                                                         const int exchange_lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float s_fused[];
  float* s_trace = s_fused;
  float* s_row = s_fused + uniform_lds_floats;
  const int p = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const bool consumer = threadIdx.x < kWave;   // wave 0 (so that the tail's threadIdx.x are its lanes)
  const int n = a.n;
  float* s_centre = s_row + n * kCoefS;
  float* s_ref = s_centre + 2 * n;
  float* s_extra = s_ref + 2 * n;    // candidate 2's controls (smp.u_extra: the LQ plan), when given
  float* s_weight = s_extra + 2 * n;
  float* s_exchange = s_fused + exchange_lds_floats + lane;   // [buffer][step of the chunk][v | kappa][lane]
  constexpr int kStepFloats = kPairValues * kWave, kBufferFloats = kPairChunk * kStepFloats;
  const int c = blockIdx.x * kWave + lane;
  const bool active = c < a.N;
  const Weights w = a.w;

  if (!consumer) {
    // ---- producer: requests, draws, the uniform operands of ITS half into LDS, then the controls chunk by chunk ----
    const SampleSpec sp = smp.spec;
    const float* __restrict__ centre = smp.centre + static_cast<size_t>(p) * smp.centre_stride;
    const float* __restrict__ ref = (smp.u_ref != nullptr) ? smp.u_ref + static_cast<size_t>(p) * n * 2 : nullptr;
    const float* __restrict__ extra = holds_candidate_2(a, smp) ? smp.u_extra + static_cast<size_t>(p) * n * 2 : nullptr;
    const float* __restrict__ knot_weight = sp.segments;
    const bool chained = smp.prev_keys != nullptr;
    constexpr int kPairs = (kStagedSteps + kWave - 1) / kWave;
    f32x2 g_centre[kPairs], g_ref[kPairs], g_extra[kPairs], g_weight[kPairs];
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      const int j = min(lane + q * kWave, n - 1);
      if (!chained) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[j];
      if (ref != nullptr) g_ref[q] = reinterpret_cast<const f32x2*>(ref)[j];
      if (extra != nullptr) g_extra[q] = reinterpret_cast<const f32x2*>(extra)[j];
      g_weight[q] = reinterpret_cast<const f32x2*>(knot_weight)[j];
    }
    int64_t prev_key[kChainBlocks / kWave];
    if (chained) {
#pragma unroll
      for (int q = 0; q < kChainBlocks / kWave; ++q)
        prev_key[q] = smp.prev_keys[static_cast<size_t>(p) * smp.prev_blocks + min(lane + q * kWave, smp.prev_blocks - 1)];
    }
    float z[kKnots][2] = {};
    const uint32_t gidx = static_cast<uint32_t>(a.index_offset + c);
    draw_normals<0, kKnots / 4>(sp, gidx, static_cast<uint32_t>(p), z);
    if (chained) {
      int64_t best = kKeyMax;
#pragma unroll
      for (int q = 0; q < kChainBlocks / kWave; ++q) {
        const int64_t kb = (lane + q * kWave < smp.prev_blocks) ? prev_key[q] : kKeyMax;
        best = (kb < best) ? kb : best;
      }
      const int64_t winner = wave_min_key(best);
      const int block = static_cast<int>(static_cast<int64_t>(static_cast<uint32_t>(winner & 0xffffffffLL)) - a.index_offset) / kWave;
      centre = smp.prev_trace + (static_cast<size_t>(p) * smp.prev_blocks + block) * smp.prev_pitch;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[min(lane + q * kWave, n - 1)];
    }
    draw_normals<kKnots / 4, kKnots / 2>(sp, gidx, static_cast<uint32_t>(p), z);
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      const int j = lane + q * kWave;
      if (j < n) {
        reinterpret_cast<f32x2*>(s_centre)[j] = g_centre[q];
        reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? g_ref[q] : g_centre[q];
        if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = g_extra[q];
        s_weight[j] = g_weight[q][1];
      }
    }
    for (int j = lane + kPairs * kWave; j < n; j += kWave) {   // horizons beyond the registers' share
      const f32x2 cj = reinterpret_cast<const f32x2*>(centre)[j];
      reinterpret_cast<f32x2*>(s_centre)[j] = cj;
      reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? reinterpret_cast<const f32x2*>(ref)[j] : cj;
      if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = reinterpret_cast<const f32x2*>(extra)[j];
      s_weight[j] = knot_weight[2 * j + 1];
    }
    __syncthreads();   // (0) the consumer has put the table rows in, this wave the rest
    // candidate 1 = the reference controls, candidate 2 = `u_extra` (the LQ plan), each when given: amplitude 0, own centre
    const bool use_extra = (gidx == 2u) && (smp.u_extra != nullptr);
    const bool use_ref = ((gidx == 1u) && (smp.u_ref != nullptr)) || use_extra;
    const f32x2* s_alt = reinterpret_cast<const f32x2*>(use_extra ? s_extra : s_ref);   // (per lane)
    const float amp = use_ref ? 0.0f : candidate_amplitude(gidx);
    struct Operands {
      f32x2 centre, ref;
      float weight;
    };
    auto request = [&](int i) {   // (one step past the end is read - inside the block - and never used)
      Operands o;
      o.centre = reinterpret_cast<const f32x2*>(s_centre)[i];
      o.ref = s_alt[i];
      o.weight = s_weight[i];
      return o;
    };
    Operands now = request(sp.knot_begin[0]);
    float* out = s_exchange;          // slot of the step being produced
    float* u_out = s_trace + lane;    // its place in the trace (rows 2i, 2i + 1)
    int in_chunk = 0, buffer = 0;
#pragma unroll
    for (int knot = 0; knot < kKnots - 1; ++knot) {
      for (int i = sp.knot_begin[knot]; i < sp.knot_begin[knot + 1]; ++i) {
        const Operands next = request(i + 1);
        float v, k;
        const float cv = use_ref ? now.ref[0] : now.centre[0];
        const float ck = use_ref ? now.ref[1] : now.centre[1];
        blend_control(sp, amp, now.weight, cv, ck, z[knot][0], z[knot][1], z[knot + 1][0], z[knot + 1][1], v, k);
        out[0] = v;
        out[kWave] = k;
        u_out[0] = v;
        u_out[kWave] = k;
        now = next;
        out += kStepFloats;
        u_out += 2 * kWave;
        if (++in_chunk == kPairChunk || i == n - 1) {   // chunk handed over (wave-uniform)
          __syncthreads();
          in_chunk = 0;
          buffer ^= 1;
          out = s_exchange + buffer * kBufferFloats;
        }
      }
    }
    return;
  }

  // ---- consumer: table rows into LDS, then the rollout one chunk behind the producer ----
  {
    const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kCoefS;
    for (int q = lane; q < n * (kCoefS / 4); q += kWave)
      reinterpret_cast<f32x4*>(s_row)[q] = reinterpret_cast<const f32x4*>(coef)[q];
  }
  const float* __restrict__ x0 = a.x0 + p * 3;
  StateS st{x0[0], x0[1], x0[2], 0.0f, 0.0f};
  __syncthreads();   // (0)
  struct StepInputs {
    float v, k;
    f32x4 lo, hi;
    float last;
  };
  auto inputs_of = [&](const float* in, int i) {
    StepInputs s;
    s.v = in[0];
    s.k = in[kWave];
    const float* t = s_row + i * kCoefS;
    s.lo = *reinterpret_cast<const f32x4*>(t);
    s.hi = *reinterpret_cast<const f32x4*>(t + 4);
    s.last = t[8];
    return s;
  };
  auto advance = [&](int i, const StepInputs& in) {
    const float row[9] = {in.lo[0], in.lo[1], in.lo[2], in.lo[3], in.hi[0], in.hi[1], in.hi[2], in.hi[3], in.last};
    step_spatial<float>(st, row, in.v, in.k, w);
    s_trace[(2 * n + 3 * i) * kWave + lane] = st.ey;
    s_trace[(2 * n + 3 * i + 1) * kWave + lane] = st.ep;
    s_trace[(2 * n + 3 * i + 2) * kWave + lane] = st.t;
  };
  for (int first = 0, buffer = 0; first < n; first += kPairChunk, buffer ^= 1) {
    __syncthreads();   // this chunk is in its buffer (and the producer may start on the other one)
    const float* in = s_exchange + buffer * kBufferFloats;
    if (first + kPairChunk <= n) {   // a full chunk, straight-line: every step's reads can move ahead of the arithmetic
      StepInputs steps[kPairChunk];
#pragma unroll
      for (int j = 0; j < kPairChunk; ++j) steps[j] = inputs_of(in + j * kStepFloats, first + j);
#pragma unroll
      for (int j = 0; j < kPairChunk; ++j) advance(first + j, steps[j]);
    } else {
      for (int j = 0; first + j < n; ++j) advance(first + j, inputs_of(in + j * kStepFloats, first + j));
    }
  }
  const float cost = finish_spatial<float>(st, w);
  s_trace[(5 * n) * kWave + lane] = st.V;
  s_trace[(5 * n + 1) * kWave + lane] = cost;
  if (active && a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c] = cost;
  const int64_t own_key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c)) : kKeyMax;
  int nfeas = (active && st.V == 0.0f) ? 1 : 0;
  int best_lane;
  const int64_t key = wave_min_key_by_lane(own_key, best_lane);   // (the index rises with the lane)
  nfeas = wave_sum_int(nfeas);
  {
    __syncthreads();   // this wave alone by now: orders its column writes before the row reads
    float* trace_out = fused.trace + (static_cast<size_t>(p) * gridDim.x + blockIdx.x) * fused.trace_pitch;
    for (int e = lane; e < 5 * n + 2; e += kWave) publish(&trace_out[e], s_trace[e * kWave + best_lane]);
  }
  if (lane == 0) {
    const size_t slot = static_cast<size_t>(p) * gridDim.x + blockIdx.x;
    publish(&a.partial_keys[slot], key);
    publish(&a.partial_feas[slot], nfeas);
  }
  fused_tail<0>(a, smp, fused, true, nullptr);
}

// The traced mode-S round on FOUR waves per workgroup.  The two-wave form above splits drawing from rolling.  Measured with
// either side stubbed out, its pace (0.144 us per step) was the DRAWING wave's blend - 124 ns per step - not the rolling
// wave's.  Of a mode S step only the 9-operation recurrence is sequential: the blend of step i needs nothing of step
// i - 1, and the bound violations read the state AFTER the step and the controls and feed nothing back (J never reads V,
// V never reads J: the split of rollout_solo_kernel).  So, one wave per SIMD of the CU:
//   waves 2, 3 (controls):       blend and leave (v, kappa) of every step in the lane's trace column - the even steps one
//                                wave, the odd steps the other;
//   wave 1 (recurrence + cost):  step_spatial_cost on those controls, one chunk behind: J, and the states into the trace;
//   wave 0 (bounds):             V += the four hinges from the trace, two chunks behind; takes J over at the end,
//                                finishes the cost, reduces, publishes and runs the tail.
// Every accumulation sees the operands of step_spatial() in its order: same bits (test_tick_forms_agree).  The Philox
// draws - a quarter of the normals each - are shared out over the four waves and exchanged through LDS, so the round's
// fixed part shrinks too.  A software pipeline over chunks of kQuadChunk steps, one workgroup barrier per chunk; every
// wave's chunk is straight-line code with all its LDS reads in front of the arithmetic.  (A fifth wave - cost and
// recurrence apart - shares a SIMD with another and set the pace there: 10.2 us per round against this form's.)
// LDS: [trace [5n + 2][64]] [table rows n x 12 | centre | reference | knot weights] [normals 16 x 64] [J 64].
#ifndef ACMPC_QUAD_CHUNK
#define ACMPC_QUAD_CHUNK 7
#endif
constexpr int kQuadChunk = ACMPC_QUAD_CHUNK;

constexpr int kQuadWaves = 4;   // bounds; recurrence + cost; two control waves

__global__ void __launch_bounds__(kQuadWaves * kWave) rollout_sampled_quad_kernel(const RolloutArgs a, const SampleArgs smp,
                                                                         const FusedFinalize fused, const int uniform_lds_floats,
                                                                         const int normals_lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float s_fused[];
  float* s_trace = s_fused;
  float* s_row = s_fused + uniform_lds_floats;
  const int p = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
  const int n = a.n;
  float* s_centre = s_row + n * kCoefS;
  float* s_ref = s_centre + 2 * n;
  float* s_extra = s_ref + 2 * n;    // candidate 2's controls (smp.u_extra: the LQ plan), when given
  float* s_weight = s_extra + 2 * n;
  float* s_z = s_fused + normals_lds_floats;          // [knot][component][lane]
  float* s_j = s_z + 2 * kKnots * kWave;
  const int c = blockIdx.x * kWave + lane;
  const bool active = c < a.N;
  const Weights w = a.w;
  const SampleSpec sp = smp.spec;
  const uint32_t gidx = static_cast<uint32_t>(a.index_offset + c);
  const int chunks = (n + kQuadChunk - 1) / kQuadChunk;
  float* col = s_trace + lane;   // rows 2i, 2i + 1 = (v, kappa) of step i; 2n + 3i .. + 2 = state after step i
  const float* __restrict__ x0 = a.x0 + p * 3;

  // this wave's quarter of the candidate's normals (Philox call `wave`: knots 2 wave, 2 wave + 1) -> LDS
  auto draw_quarter = [&](auto quarter) {
    constexpr int Q = decltype(quarter)::value;
    float z[kKnots][2] = {};
    draw_normals<Q, Q + 1>(sp, gidx, static_cast<uint32_t>(p), z);
#pragma unroll
    for (int knot = 2 * Q; knot < 2 * Q + 2; ++knot) {
      s_z[(2 * knot) * kWave + lane] = z[knot][0];
      s_z[(2 * knot + 1) * kWave + lane] = z[knot][1];
    }
  };
  static_assert(kKnots == 8, "four Philox calls per candidate, one per wave");

  // the controls of the steps i = parity, parity + 2, ... (every step with one control wave's worth of work halved): blend,
  // leave (v, kappa) in the trace; one barrier per chunk, taken by both control waves
  auto produce_controls = [&](const float (&z)[kKnots][2], const int parity) {
    // candidate 1 = the reference controls, candidate 2 = `u_extra` (the LQ plan), each when given: amplitude 0, own centre
    const bool use_extra = (gidx == 2u) && (smp.u_extra != nullptr);
    const bool use_ref = ((gidx == 1u) && (smp.u_ref != nullptr)) || use_extra;
    const f32x2* s_alt = reinterpret_cast<const f32x2*>(use_extra ? s_extra : s_ref);   // (per lane)
    const float amp = use_ref ? 0.0f : candidate_amplitude(gidx);
    struct Operands {
      f32x2 centre, ref;
      float weight;
    };
    auto request = [&](int i) {   // (up to two steps past the end are read - inside the block's padding - and never used)
      Operands o;
      o.centre = reinterpret_cast<const f32x2*>(s_centre)[i];
      o.ref = s_alt[i];
      o.weight = s_weight[i];
      return o;
    };
    // (Broadcasting the operands out of the registers they were fetched into with v_readlane, instead of reading them
    // back from LDS one step ahead: 187 ns per step against 144 - measured, not kept.)
    Operands now = request(parity);
    int in_chunk = 0;
#pragma unroll
    for (int knot = 0; knot < kKnots - 1; ++knot) {
      for (int i = sp.knot_begin[knot]; i < sp.knot_begin[knot + 1]; ++i) {
        if ((i & 1) == parity) {   // (wave-uniform)
          const Operands next = request(i + 2);   // one own step ahead of the arithmetic
          float v, k;
          blend_control(sp, amp, now.weight, use_ref ? now.ref[0] : now.centre[0], use_ref ? now.ref[1] : now.centre[1],
                        z[knot][0], z[knot][1], z[knot + 1][0], z[knot + 1][1], v, k);
          col[(2 * i) * kWave] = v;
          col[(2 * i + 1) * kWave] = k;
          now = next;
        }
        if (++in_chunk == kQuadChunk || i == n - 1) {   // chunk handed on (wave-uniform)
          __syncthreads();
          in_chunk = 0;
        }
      }
    }
  };

  if (wave == 3) {
    // ---- the second control wave: its quarter of the draws, nothing to stage; the odd steps of every chunk ----
    draw_quarter(std::integral_constant<int, 3>{});
    __syncthreads();   // (S)
    float z[kKnots][2];
#pragma unroll
    for (int knot = 0; knot < kKnots; ++knot) {
      z[knot][0] = s_z[(2 * knot) * kWave + lane];
      z[knot][1] = s_z[(2 * knot + 1) * kWave + lane];
    }
    produce_controls(z, 1);
    return;
  }
  if (wave == 2) {
    // ---- controls: requests, its quarter of the draws, the uniform operands of the control waves into LDS, then chunk by chunk ----
    const float* __restrict__ centre = smp.centre + static_cast<size_t>(p) * smp.centre_stride;
    const float* __restrict__ ref = (smp.u_ref != nullptr) ? smp.u_ref + static_cast<size_t>(p) * n * 2 : nullptr;
    const float* __restrict__ extra = holds_candidate_2(a, smp) ? smp.u_extra + static_cast<size_t>(p) * n * 2 : nullptr;
    const float* __restrict__ knot_weight = sp.segments;
    const bool chained = smp.prev_keys != nullptr;
    constexpr int kPairs = (kStagedSteps + kWave - 1) / kWave;
    f32x2 g_centre[kPairs], g_ref[kPairs], g_extra[kPairs], g_weight[kPairs];
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      const int j = min(lane + q * kWave, n - 1);
      if (!chained) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[j];
      if (ref != nullptr) g_ref[q] = reinterpret_cast<const f32x2*>(ref)[j];
      if (extra != nullptr) g_extra[q] = reinterpret_cast<const f32x2*>(extra)[j];
      g_weight[q] = reinterpret_cast<const f32x2*>(knot_weight)[j];
    }
    int64_t prev_key[kChainBlocks / kWave];
    if (chained) {
#pragma unroll
      for (int q = 0; q < kChainBlocks / kWave; ++q)
        prev_key[q] = smp.prev_keys[static_cast<size_t>(p) * smp.prev_blocks + min(lane + q * kWave, smp.prev_blocks - 1)];
    }
    draw_quarter(std::integral_constant<int, 2>{});
    if (chained) {
      int64_t best = kKeyMax;
#pragma unroll
      for (int q = 0; q < kChainBlocks / kWave; ++q) {
        const int64_t kb = (lane + q * kWave < smp.prev_blocks) ? prev_key[q] : kKeyMax;
        best = (kb < best) ? kb : best;
      }
      const int64_t winner = wave_min_key(best);
      const int block = static_cast<int>(static_cast<int64_t>(static_cast<uint32_t>(winner & 0xffffffffLL)) - a.index_offset) / kWave;
      centre = smp.prev_trace + (static_cast<size_t>(p) * smp.prev_blocks + block) * smp.prev_pitch;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[min(lane + q * kWave, n - 1)];
    }
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      const int j = lane + q * kWave;
      if (j < n) {
        reinterpret_cast<f32x2*>(s_centre)[j] = g_centre[q];
        reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? g_ref[q] : g_centre[q];
        if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = g_extra[q];
        s_weight[j] = g_weight[q][1];
      }
    }
    for (int j = lane + kPairs * kWave; j < n; j += kWave) {   // horizons beyond the registers' share
      const f32x2 cj = reinterpret_cast<const f32x2*>(centre)[j];
      reinterpret_cast<f32x2*>(s_centre)[j] = cj;
      reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? reinterpret_cast<const f32x2*>(ref)[j] : cj;
      if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = reinterpret_cast<const f32x2*>(extra)[j];
      s_weight[j] = knot_weight[2 * j + 1];
    }
    __syncthreads();   // (S) every wave's normals and the table rows are in
    float z[kKnots][2];
#pragma unroll
    for (int knot = 0; knot < kKnots; ++knot) {
      z[knot][0] = s_z[(2 * knot) * kWave + lane];
      z[knot][1] = s_z[(2 * knot + 1) * kWave + lane];
    }
    produce_controls(z, 0);
    return;
  }

  // ---- the other two waves: table rows into LDS, their quarters of the draws ----
  {
    const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kCoefS;
    for (int q = threadIdx.x; q < n * (kCoefS / 4); q += 2 * kWave)
      reinterpret_cast<f32x4*>(s_row)[q] = reinterpret_cast<const f32x4*>(coef)[q];
  }
  if (wave == 0) draw_quarter(std::integral_constant<int, 0>{});
  if (wave == 1) draw_quarter(std::integral_constant<int, 1>{});
  __syncthreads();   // (S)

  if (wave == 1) {
    // ---- recurrence + cost: one chunk behind the controls ----
    StateS st{x0[0], x0[1], x0[2], 0.0f, 0.0f};
    auto roll = [&](int i, float v, float k, const f32x4& lo, const f32x4& hi) {
      const float cr[7] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2]};
      step_spatial_cost(st, cr, v, k, w);
      col[(2 * n + 3 * i) * kWave] = st.ey;
      col[(2 * n + 3 * i + 1) * kWave] = st.ep;
      col[(2 * n + 3 * i + 2) * kWave] = st.t;
    };
    __syncthreads();   // chunk 0 of the controls is in
    for (int t = 1; t <= chunks; ++t) {
      const int first = (t - 1) * kQuadChunk;
      if (first + kQuadChunk <= n) {   // a full chunk, straight-line: every step's reads move ahead of the arithmetic
        float v[kQuadChunk], k[kQuadChunk];
        f32x4 lo[kQuadChunk], hi[kQuadChunk];
#pragma unroll
        for (int q = 0; q < kQuadChunk; ++q) {
          v[q] = col[(2 * (first + q)) * kWave];
          k[q] = col[(2 * (first + q) + 1) * kWave];
          lo[q] = *reinterpret_cast<const f32x4*>(s_row + (first + q) * kCoefS);
          hi[q] = *reinterpret_cast<const f32x4*>(s_row + (first + q) * kCoefS + 4);
        }
#pragma unroll
        for (int q = 0; q < kQuadChunk; ++q) roll(first + q, v[q], k[q], lo[q], hi[q]);
      } else {
        for (int i = first; i < n; ++i)
          roll(i, col[(2 * i) * kWave], col[(2 * i + 1) * kWave], *reinterpret_cast<const f32x4*>(s_row + i * kCoefS),
               *reinterpret_cast<const f32x4*>(s_row + i * kCoefS + 4));
      }
      if (t == chunks) s_j[lane] = st.J;   // (ordered before the bounds wave's read by the two barriers that follow there)
      __syncthreads();   // chunk t - 1 rolled (and, while t < chunks, chunk t of the controls is in)
    }
    return;
  }

  // ---- bounds (wave 0): two chunks behind the controls ----
  StateS st{x0[0], x0[1], x0[2], 0.0f, 0.0f};
  __syncthreads();   // chunk time 0
  __syncthreads();   // chunk time 1: chunk 0 rolled
  {
    auto account = [&](float v, float k, float ey_lo, float ey_hi, float ey_after, float t_after) {
      st.V = st.V + hinge2<float>(w.ulo0 - v, v - w.uhi0);
      st.V = st.V + hinge2<float>(w.ulo1 - k, k - w.uhi1);
      st.V = st.V + hinge2<float>(ey_lo - ey_after, ey_after - ey_hi);
      const float tv = vmax(w.tmin - t_after, 0.0f);
      st.V = st.V + tv * tv;
    };
    for (int t = 2; t <= chunks + 1; ++t) {
      const int first = (t - 2) * kQuadChunk;
      if (first + kQuadChunk <= n) {
        float v[kQuadChunk], k[kQuadChunk], ey[kQuadChunk], tt[kQuadChunk], lo[kQuadChunk], hi[kQuadChunk];
#pragma unroll
        for (int q = 0; q < kQuadChunk; ++q) {
          const int i = first + q;
          v[q] = col[(2 * i) * kWave];
          k[q] = col[(2 * i + 1) * kWave];
          ey[q] = col[(2 * n + 3 * i) * kWave];
          tt[q] = col[(2 * n + 3 * i + 2) * kWave];
          lo[q] = s_row[i * kCoefS + 7];
          hi[q] = s_row[i * kCoefS + 8];
        }
#pragma unroll
        for (int q = 0; q < kQuadChunk; ++q) account(v[q], k[q], lo[q], hi[q], ey[q], tt[q]);
      } else {
        for (int i = first; i < n; ++i)
          account(col[(2 * i) * kWave], col[(2 * i + 1) * kWave], s_row[i * kCoefS + 7], s_row[i * kCoefS + 8],
                  col[(2 * n + 3 * i) * kWave], col[(2 * n + 3 * i + 2) * kWave]);
      }
      __syncthreads();   // (the last one: J handed over)
    }
    // the state after the last step, for the terminal cost
    st.ey = col[(2 * n + 3 * (n - 1)) * kWave];
    st.ep = col[(2 * n + 3 * (n - 1) + 1) * kWave];
    st.t = col[(2 * n + 3 * (n - 1) + 2) * kWave];
  }
  st.J = s_j[lane];
  const float cost = finish_spatial<float>(st, w);   // (st.ey / ep / t: the state after the last step)
  s_trace[(5 * n) * kWave + lane] = st.V;
  s_trace[(5 * n + 1) * kWave + lane] = cost;
  if (active && a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c] = cost;
  const int64_t own_key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c)) : kKeyMax;
  int best_lane;
  const int64_t key = wave_min_key_by_lane(own_key, best_lane);   // (the index rises with the lane)
  const int nfeas = wave_sum_int((active && st.V == 0.0f) ? 1 : 0);
  {
    __syncthreads();   // this wave alone by now: orders its column writes before the row reads
    float* trace_out = fused.trace + (static_cast<size_t>(p) * gridDim.x + blockIdx.x) * fused.trace_pitch;
    for (int e = lane; e < 5 * n + 2; e += kWave) publish(&trace_out[e], s_trace[e * kWave + best_lane]);
  }
  if (lane == 0) {
    const size_t slot = static_cast<size_t>(p) * gridDim.x + blockIdx.x;
    publish(&a.partial_keys[slot], key);
    publish(&a.partial_feas[slot], nfeas);
  }
  fused_tail<0>(a, smp, fused, true, nullptr);
}

// The traced mode-T round on THREE waves per workgroup.  On one wave a mode T step is a serial stream of ~110 instructions
// with two dependent LDS gathers in it (the window's keys, then the nearest waypoint's row): 0.6 us per step, 29 us per
// round.  But the pose does not depend on the search - only the cost does - so the step falls into three stages that
// meet in LDS, in the trace the round keeps anyway:
//   wave 2 (poses):  draws and blends the controls, integrates the pose (temporal_advance), leaves (v, kappa) and
//                    (X, Y, phi) of every step in the lane's trace column;
//   wave 1 (search): reads (X, Y), finds the nearest waypoint from the previous step's (the one sequential chain left),
//                    leaves the index;
//   wave 0 (costs):  reads pose, controls and index, gathers the waypoint's row, accumulates the cost terms
//                    (temporal_cost) - independent from step to step, so its gathers overlap - then reduces, publishes
//                    and runs the tail as the single-wave kernel does.
// A software pipeline over chunks of kTrioChunk steps, one workgroup barrier per chunk: at chunk time t wave 2 works on
// chunk t, wave 1 on chunk t - 1, wave 0 on chunk t - 2.  The search wave sets the pace (~0.2 us per step: its keys are a
// dependent LDS gather per step), so the chunk only decides how long the pipeline takes to fill and drain: measured tick
// p50 at chunks of 10 / 7 / 4 / 3 / 2 steps: 67.1 / 65.5 / 63.0 / 61.8 / 62.7 us (one wave per workgroup: 86.3).  Same operations on the same operands: same bits
// (test_tick_forms_agree with ACMPC_NO_TRIO_ROUNDS).
// LDS: [waypoint rows n x 8 | key table n x 3] [trace [5n + 2][64]] [uniform operands: centre, reference, knot weights]
//      [nearest indices [n][64], 16-bit].
#ifndef ACMPC_TRIO_CHUNK
#define ACMPC_TRIO_CHUNK 3
#endif
constexpr int kTrioChunk = ACMPC_TRIO_CHUNK;

__global__ void __launch_bounds__(3 * kWave) rollout_sampled_trio_kernel(const RolloutArgs a, const SampleArgs smp,
                                                                         const FusedFinalize fused, const int trace_lds_floats,
                                                                         const int uniform_lds_floats,
                                                                         const int index_lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float s_fused[];
  const int p = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
  const int n = a.n;
  const Weights w = a.w;
  float* s_wp = s_fused;
  float* s_abc = s_wp + n * kCoefT;
  float* s_trace = s_fused + trace_lds_floats;
  float* s_centre = s_fused + uniform_lds_floats;
  float* s_ref = s_centre + 2 * n;
  float* s_extra = s_ref + 2 * n;    // candidate 2's controls (smp.u_extra: the LQ plan), when given
  float* s_weight = s_extra + 2 * n;
  // (16-bit entries: horizons go to 1 025 steps; as 32-bit ones the frames of the verified search stopped fitting beside the
  // trace at the mapping controller's horizon of 100 once the key table grew to 32 bytes per waypoint)
  unsigned short* s_index = reinterpret_cast<unsigned short*>(s_fused + index_lds_floats);
  const int c = blockIdx.x * kWave + lane;
  const bool active = c < a.N;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kCoefT;
  const float* __restrict__ x0 = a.x0 + p * 3;
  const int chunks = (n + kTrioChunk - 1) / kTrioChunk;
  float* col = s_trace + lane;   // this lane's trace column: rows 2i, 2i + 1 = (v, kappa); 2n + 3i .. + 2 = pose after step i

  if (wave == 2) {
    // ---- poses: requests, draws, uniform operands into LDS, then controls and poses chunk by chunk ----
    const SampleSpec sp = smp.spec;
    const float* __restrict__ centre = smp.centre + static_cast<size_t>(p) * smp.centre_stride;
    const float* __restrict__ ref = (smp.u_ref != nullptr) ? smp.u_ref + static_cast<size_t>(p) * n * 2 : nullptr;
    const float* __restrict__ extra = holds_candidate_2(a, smp) ? smp.u_extra + static_cast<size_t>(p) * n * 2 : nullptr;
    const float* __restrict__ knot_weight = sp.segments;
    const bool chained = smp.prev_keys != nullptr;
    constexpr int kPairs = (kStagedSteps + kWave - 1) / kWave;
    f32x2 g_centre[kPairs], g_ref[kPairs], g_extra[kPairs], g_weight[kPairs];
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      const int j = min(lane + q * kWave, n - 1);
      if (!chained) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[j];
      if (ref != nullptr) g_ref[q] = reinterpret_cast<const f32x2*>(ref)[j];
      if (extra != nullptr) g_extra[q] = reinterpret_cast<const f32x2*>(extra)[j];
      g_weight[q] = reinterpret_cast<const f32x2*>(knot_weight)[j];
    }
    int64_t prev_key[kChainBlocks / kWave];
    if (chained) {
#pragma unroll
      for (int q = 0; q < kChainBlocks / kWave; ++q)
        prev_key[q] = smp.prev_keys[static_cast<size_t>(p) * smp.prev_blocks + min(lane + q * kWave, smp.prev_blocks - 1)];
    }
    float z[kKnots][2] = {};
    const uint32_t gidx = static_cast<uint32_t>(a.index_offset + c);
    draw_normals<0, kKnots / 4>(sp, gidx, static_cast<uint32_t>(p), z);
    if (chained) {
      int64_t best = kKeyMax;
#pragma unroll
      for (int q = 0; q < kChainBlocks / kWave; ++q) {
        const int64_t kb = (lane + q * kWave < smp.prev_blocks) ? prev_key[q] : kKeyMax;
        best = (kb < best) ? kb : best;
      }
      const int64_t winner = wave_min_key(best);
      const int block = static_cast<int>(static_cast<int64_t>(static_cast<uint32_t>(winner & 0xffffffffLL)) - a.index_offset) / kWave;
      centre = smp.prev_trace + (static_cast<size_t>(p) * smp.prev_blocks + block) * smp.prev_pitch;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) g_centre[q] = reinterpret_cast<const f32x2*>(centre)[min(lane + q * kWave, n - 1)];
    }
    draw_normals<kKnots / 4, kKnots / 2>(sp, gidx, static_cast<uint32_t>(p), z);
#pragma unroll
    for (int q = 0; q < kPairs; ++q) {
      const int j = lane + q * kWave;
      if (j < n) {
        reinterpret_cast<f32x2*>(s_centre)[j] = g_centre[q];
        reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? g_ref[q] : g_centre[q];
        if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = g_extra[q];
        s_weight[j] = g_weight[q][1];
      }
    }
    for (int j = lane + kPairs * kWave; j < n; j += kWave) {   // horizons beyond the registers' share
      const f32x2 cj = reinterpret_cast<const f32x2*>(centre)[j];
      reinterpret_cast<f32x2*>(s_centre)[j] = cj;
      reinterpret_cast<f32x2*>(s_ref)[j] = (ref != nullptr) ? reinterpret_cast<const f32x2*>(ref)[j] : cj;
      if (extra != nullptr) reinterpret_cast<f32x2*>(s_extra)[j] = reinterpret_cast<const f32x2*>(extra)[j];
      s_weight[j] = knot_weight[2 * j + 1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this wave reads what its own lanes staged
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // candidate 1 = the reference controls, candidate 2 = `u_extra` (the LQ plan), each when given: amplitude 0, own centre
    const bool use_extra = (gidx == 2u) && (smp.u_extra != nullptr);
    const bool use_ref = ((gidx == 1u) && (smp.u_ref != nullptr)) || use_extra;
    const f32x2* s_alt = reinterpret_cast<const f32x2*>(use_extra ? s_extra : s_ref);   // (per lane)
    const float amp = use_ref ? 0.0f : candidate_amplitude(gidx);
    StateT pose = start_temporal<float>(x0, coef);
    struct Operands {
      f32x2 centre, ref;
      float weight;
    };
    auto request = [&](int i) {   // (one step past the end is read - inside the block - and never used)
      Operands o;
      o.centre = reinterpret_cast<const f32x2*>(s_centre)[i];
      o.ref = s_alt[i];
      o.weight = s_weight[i];
      return o;
    };
    Operands now = request(sp.knot_begin[0]);
    int in_chunk = 0;
#pragma unroll
    for (int knot = 0; knot < kKnots - 1; ++knot) {
      for (int i = sp.knot_begin[knot]; i < sp.knot_begin[knot + 1]; ++i) {
        const Operands next = request(i + 1);
        float v, k;
        blend_control(sp, amp, now.weight, use_ref ? now.ref[0] : now.centre[0], use_ref ? now.ref[1] : now.centre[1],
                      z[knot][0], z[knot][1], z[knot + 1][0], z[knot + 1][1], v, k);
        now = next;
        temporal_advance<float>(pose, v, k, w);
        col[(2 * i) * kWave] = v;
        col[(2 * i + 1) * kWave] = k;
        col[(2 * n + 3 * i) * kWave] = pose.X;
        col[(2 * n + 3 * i + 1) * kWave] = pose.Y;
        col[(2 * n + 3 * i + 2) * kWave] = pose.phi;
        if (++in_chunk == kTrioChunk || i == n - 1) {   // chunk handed on (wave-uniform)
          __syncthreads();
          in_chunk = 0;
        }
      }
    }
    return;
  }

  // ---- the other two waves put the waypoint tables into LDS (the pose wave does not read them) ----
  stage_temporal_tables(coef, n, static_cast<int>(threadIdx.x), 2 * kWave, s_wp, s_abc);
  float* s_frames = s_wp + ((n * (kCoefT + kKeyStride) + 3) & ~3);   // exhaustive search: its frames, when given
  if (a.nn_frames != nullptr) {
    const float* __restrict__ frames = a.nn_frames + static_cast<size_t>(p) * verified_frame_floats(n);
    for (int e = static_cast<int>(threadIdx.x); e < verified_frame_floats(n); e += 2 * kWave) s_frames[e] = frames[e];
  }
  if (wave == 1) {
    // ---- search: one chunk behind the poses ----
    int j_prev = 0;
    __syncthreads();   // chunk 0 of the poses is in (and - a barrier orders all LDS writes before it - the tables)
    with_search_kind(w, n, [&](auto kind) {
      for (int t = 1; t <= chunks; ++t) {
        const int first = (t - 1) * kTrioChunk;
        // the chunk's positions first (they do not depend on the search): one round trip, not one per step
        float X[kTrioChunk], Y[kTrioChunk];
#pragma unroll
        for (int q = 0; q < kTrioChunk; ++q) {
          const int i = min(first + q, n - 1);
          X[q] = col[(2 * n + 3 * i) * kWave];
          Y[q] = col[(2 * n + 3 * i + 1) * kWave];
        }
#pragma unroll
        for (int q = 0; q < kTrioChunk; ++q) {
          if (first + q < n) {   // (wave-uniform)
            if constexpr (decltype(kind)::value == kSearchVerified) {
              j_prev = propose_nearest(X[q], Y[q], s_abc, n, j_prev);   // (confirmed by the cost wave)
            } else {
              j_prev = search_temporal_as<decltype(kind)::value>(X[q], Y[q], s_abc, n, w, j_prev);
            }
            s_index[(first + q) * kWave + lane] = static_cast<unsigned short>(j_prev);
          }
        }
        __syncthreads();   // chunk t - 1 searched (and, while t < chunks, chunk t of the poses is in)
      }
    }, a.nn_frames != nullptr);
    return;
  }

  // ---- costs: two chunks behind the poses ----
  const bool confirm = search_kind(w, n, a.nn_frames != nullptr) == kSearchVerified;
  StateT st = start_temporal<float>(x0, coef);
  __syncthreads();   // chunk time 0
  __syncthreads();   // chunk time 1: chunk 0 searched
  for (int t = 2; t <= chunks + 1; ++t) {
    const int last = min((t - 1) * kTrioChunk, n);
#pragma unroll
    for (int q = 0; q < kTrioChunk; ++q) {
      const int i = (t - 2) * kTrioChunk + q;
      if (i < last) {   // (wave-uniform)
        st.X = col[(2 * n + 3 * i) * kWave];
        st.Y = col[(2 * n + 3 * i + 1) * kWave];
        st.phi = col[(2 * n + 3 * i + 2) * kWave];
        int j = s_index[i * kWave + lane];
        if (confirm) {   // (wave-uniform) the search wave's proposal, held against the frame of its window
          const int before = (i == 0) ? 0 : s_index[(i - 1) * kWave + lane];
          j = confirm_nearest(st.X, st.Y, s_abc, s_frames, n, before, j);
        }
        temporal_cost(st, s_wp + j * kCoefT, col[(2 * i) * kWave], col[(2 * i + 1) * kWave], w);
      }
    }
    if (t <= chunks) __syncthreads();   // (the last chunk times have no partner left to wait for)
  }
  const float cost = finish_temporal<float>(st, n, w);
  s_trace[(5 * n) * kWave + lane] = st.V;
  s_trace[(5 * n + 1) * kWave + lane] = cost;
  if (active && a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c] = cost;
  const int64_t own_key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c)) : kKeyMax;
  int best_lane;
  const int64_t key = wave_min_key_by_lane(own_key, best_lane);   // (the index rises with the lane)
  const int nfeas = wave_sum_int((active && st.V == 0.0f) ? 1 : 0);
  {
    __syncthreads();   // this wave alone by now: orders its column writes before the row reads
    float* trace_out = fused.trace + (static_cast<size_t>(p) * gridDim.x + blockIdx.x) * fused.trace_pitch;
    // (the waves exchanged the poses in the path's own frame - start_temporal() - and they leave in the caller's: rows
    // 2n + 3i and 2n + 3i + 1 of the trace are X and Y after step i)
    const float ox = coef[0], oy = coef[1];
    for (int e = lane; e < 5 * n + 2; e += kWave) {
      const int k = e - 2 * n;
      const bool is_x = k >= 0 && k < 3 * n && k % 3 == 0, is_y = k >= 0 && k < 3 * n && k % 3 == 1;
      const float value = s_trace[e * kWave + best_lane];
      publish(&trace_out[e], is_x ? value + ox : is_y ? value + oy : value);   // (the same add as every other form's)
    }
  }
  if (lane == 0) {
    const size_t slot = static_cast<size_t>(p) * gridDim.x + blockIdx.x;
    publish(&a.partial_keys[slot], key);
    publish(&a.partial_feas[slot], nfeas);
  }
  fused_tail<1>(a, smp, fused, true, nullptr);
}

// ---- one problem (or a few) of a few thousand candidates per call: ONE launch, the winner never rolled twice ------
// acmpc_solve_device on a caller's control matrix, mode S.  What rollout_kernel + finalize_kernel do in two launches -
// the second re-rolling the winner on one wave, which takes as long as the rollout itself at this size - is one launch
// here: 64 candidates per workgroup; every lane leaves the STATES of its candidate in its row of an LDS block
// [64][3n | 1] (an odd pitch: the lanes' writes of one step fall on 64 different banks, and a row reads back
// contiguously; 38 kB at H = 50: four workgroups per CU); the workgroup publishes the row of its best candidate
// (+ violation and cost) as its trace; the workgroup that finishes a problem last (tickets, as in the fused rounds)
// takes the argmin over the partial keys and assembles the record out of the winning workgroup's trace and the
// winner's row of the control matrix - copies only, so the record holds exactly the bits the winning lane computed.
//
// SPLIT: two waves per workgroup roll the same 64 candidates, wave 0 the stage cost (step_spatial_cost), wave 1 the
// bound violations (step_spatial_bounds) and the trace; V crosses once, after the horizon.  A lone wave issues one
// instruction every ~2 ns whatever it is - scalar ones included - and a second wave on a SIMD issues in the gaps of the
// first, so a launch takes as long as its longest instruction stream: ~40 instructions per step instead of ~60, also
// when the launch has two waves for every SIMD (1 024 workgroups: measured 20.4 us against 22.1).
// Candidate-major matrices (LAYOUT 0): the workgroup's 64 rows are one contiguous span, copied into LDS with 16-byte
// loads by all its waves and read back row-wise (rollout_tile_kernel's scheme).
// LDS: [64][3n | 1] states | [64] V | (LAYOUT 0) [64][2n] control tile.
// NSTEPS > 0: the horizon is the compile-time constant NSTEPS (49 = every racing configuration of the reference,
// configs/*.yaml: horizon 50) and the states stay in REGISTERS - 3 n of them, the step loop fully unrolled, each
// state computed into its final register, so the trace costs no instruction at all where the LDS form pays three
// ds_write per step (0.8 us of a 49-step walk: LDS writes share the wait counter of the scalar row loads).  Only the
// best lane's states ever reach the LDS: it dumps its registers after the reduction and the wave reads them back side
// by side.  The roles of SPLIT swap with it: wave 0 rolls the bounds and keeps the states, wave 1 the stage cost.
__device__ __forceinline__ int solo_pitch(int n) { return (3 * n) | 1; }

template <int LAYOUT, bool SPLIT, int NSTEPS>
__global__ void __launch_bounds__(SPLIT ? 2 * kWave : kWave) rollout_solo_kernel(const RolloutArgs a,
                                                                                  const FusedFinalize fused) {
  extern __shared__ __attribute__((aligned(16))) float s_solo[];
  constexpr bool kRegs = NSTEPS > 0;
  const int p = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const bool first = threadIdx.x < kWave;   // wave 0: its threadIdx.x are its lanes (the tail relies on it)
  const int n = kRegs ? NSTEPS : a.n;
  const int c = blockIdx.x * kWave + lane;
  const bool active = c < a.N;
  const int c_run = active ? c : a.N - 1;   // spare lanes of the last workgroup roll the last candidate; never reported
  const Weights w = a.w;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kCoefS;
  const float* __restrict__ x0 = a.x0 + p * 3;
  const int pitch = solo_pitch(n);
  // LDS: the states ([64][pitch], or - registers - one row of 3n) | [64] the other wave's sum | the control tile
  float* s_other = s_solo + (kRegs ? ((3 * n + 3) & ~3) : kWave * pitch);
  float* s_tile = s_other + kWave;
  ACMPC_STAMP(0);
  if constexpr (LAYOUT == 0) {
    const int c0 = blockIdx.x * kWave;
    const int rows = min(kWave, a.N - c0);
    const int total = rows * 2 * n;
    const size_t span = (static_cast<size_t>(p) * a.N + c0) * 2 * n;  // float index of the span
    const float* __restrict__ src = a.U + span;
    constexpr int kThreads = SPLIT ? 2 * kWave : kWave;
    if (((span & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.U) & 15u) == 0)) {
      const int quads = total >> 2;
      for (int q = threadIdx.x; q < quads; q += kThreads)
        reinterpret_cast<f32x4*>(s_tile)[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + q);
      for (int e = (quads << 2) + threadIdx.x; e < total; e += kThreads) s_tile[e] = src[e];
    } else {   // the span starts on an 8-byte boundary only
      for (int q = threadIdx.x; q < (total >> 1); q += kThreads)
        reinterpret_cast<f32x2*>(s_tile)[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(src) + q);
    }
    __syncthreads();
  }
  const f32x2* s_row = reinterpret_cast<const f32x2*>(s_tile + (c_run - blockIdx.x * kWave) * 2 * n);
  auto controls = [&](int i, float& v, float& k) {
    if constexpr (LAYOUT == 1) {
      float vv[1], kk[1];
      load_controls<1, 1>(a.U, p, a.N, n, i, c_run, vv, kk);
      v = vv[0];
      k = kk[0];
    } else {
      const f32x2 vk = s_row[i];
      v = vk[0];
      k = vk[1];
    }
  };
  StateS st{x0[0], x0[1], x0[2], 0.0f, 0.0f};
  // which wave keeps the trace: LDS form - wave 1 (the bounds wave, the shorter stream without it); registers - wave 0
  constexpr bool kHelperRollsCost = kRegs;   // the helper (wave 1) rolls the stage cost, else the bounds
  if (SPLIT && !first) {
    ACMPC_STAMP(1);
    float* mine = s_solo + lane * pitch;
#pragma unroll 7
    for (int i = 0; i < n; ++i) {
      float v, k;
      controls(i, v, k);
      if constexpr (kHelperRollsCost) {
        step_spatial_cost(st, coef + i * kCoefS, v, k, w);
      } else {
        step_spatial_bounds(st, coef + i * kCoefS, v, k, w);
        mine[3 * i] = st.ey;
        mine[3 * i + 1] = st.ep;
        mine[3 * i + 2] = st.t;
      }
    }
    s_other[lane] = kHelperRollsCost ? st.J : st.V;
    ACMPC_STAMP(2);
    __syncthreads();   // (1)
    return;
  }
  ACMPC_STAMP(1);
  float xs[kRegs ? NSTEPS : 1][3];
  if constexpr (kRegs) {
#pragma unroll
    for (int i = 0; i < NSTEPS; ++i) {
      float v, k;
      controls(i, v, k);
      if constexpr (SPLIT) {
        step_spatial_bounds(st, coef + i * kCoefS, v, k, w);
      } else {
        step_spatial<float>(st, coef + i * kCoefS, v, k, w);
      }
      xs[i][0] = st.ey;
      xs[i][1] = st.ep;
      xs[i][2] = st.t;
    }
  } else {
    float* mine = s_solo + lane * pitch;
#pragma unroll 7
    for (int i = 0; i < n; ++i) {
      float v, k;
      controls(i, v, k);
      if constexpr (SPLIT) {
        step_spatial_cost(st, coef + i * kCoefS, v, k, w);
      } else {
        step_spatial<float>(st, coef + i * kCoefS, v, k, w);
        mine[3 * i] = st.ey;
        mine[3 * i + 1] = st.ep;
        mine[3 * i + 2] = st.t;
      }
    }
  }
  ACMPC_STAMP(2);
  __syncthreads();   // (1) split: the other wave's sum (and trace) is in; one wave: orders its own trace writes
  if constexpr (SPLIT) {
    if constexpr (kHelperRollsCost) {
      st.J = s_other[lane];
    } else {
      st.V = s_other[lane];
    }
  }
  const float cost = finish_spatial<float>(st, w);
  if (active && a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c] = cost;
  const int64_t own_key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c)) : kKeyMax;
  int best_lane;
  const int64_t key = wave_min_key_by_lane(own_key, best_lane);   // (the index rises with the lane)
  const int nfeas = wave_sum_int((active && st.V == 0.0f) ? 1 : 0);
  const int blocks = static_cast<int>(gridDim.x);
  ACMPC_STAMP(3);
  const size_t slot = static_cast<size_t>(p) * blocks + blockIdx.x;
  if (fused.records != nullptr) {
    const float best_v = bcast(st.V, best_lane), best_cost = bcast(cost, best_lane);
    const float* row = s_solo + best_lane * pitch;
    if constexpr (kRegs) {
      row = s_solo;
      if (lane == best_lane) {
#pragma unroll
        for (int i = 0; i < NSTEPS; ++i) {
          s_solo[3 * i] = xs[i][0];
          s_solo[3 * i + 1] = xs[i][1];
          s_solo[3 * i + 2] = xs[i][2];
        }
      }
      __syncthreads();   // this wave alone by now: orders the one lane's writes before the wave's reads
    }
    float* trace_out = fused.trace + slot * fused.trace_pitch;
    for (int e = lane; e < 3 * n; e += kWave) publish(&trace_out[e], row[e]);
    if (lane == 0) {
      publish(&trace_out[3 * n], best_v);
      publish(&trace_out[3 * n + 1], best_cost);
    }
  }
  if (lane == 0) {
    publish(&a.partial_keys[slot], key);
    publish(&a.partial_feas[slot], nfeas);
  }
  ACMPC_STAMP(4);
  if (!last_workgroup_of_problem(fused.tickets + static_cast<size_t>(p) * (fused.ticket_groups + 1) * kTicketStride,
                                 fused.ticket_groups))
    return;

  // ---- the problem's last workgroup: argmin over the partial keys, record = copies ----
  constexpr int kPerLane = kSoloBlocks / kWave;   // every key requested before the first is looked at: one round trip
  int64_t kb[kPerLane];
  int fb[kPerLane];
#pragma unroll
  for (int q = 0; q < kPerLane; ++q) {
    kb[q] = kKeyMax;
    fb[q] = 0;
    if (q * kWave < blocks) {   // (wave-uniform)
      const int b = min(lane + q * kWave, blocks - 1);
      kb[q] = observe(&a.partial_keys[static_cast<size_t>(p) * blocks + b]);
      fb[q] = observe(&a.partial_feas[static_cast<size_t>(p) * blocks + b]);
    }
  }
  int64_t best = kKeyMax;
  int total_feas = 0;
#pragma unroll
  for (int q = 0; q < kPerLane; ++q) {
    if (q * kWave < blocks) {
      const bool mine_too = lane + q * kWave < blocks;   // (a clamped lane re-read the last workgroup's slot)
      best = (mine_too && kb[q] < best) ? kb[q] : best;
      total_feas += mine_too ? fb[q] : 0;
    }
  }
  total_feas = wave_sum_int(total_feas);
  // (workgroup b's candidates are b * 64 .., so the winner's workgroup follows from its index)
  const int64_t winner = wave_min_key(best);
  const uint32_t best_lo = static_cast<uint32_t>(winner & 0xffffffffLL);
  ACMPC_STAMP(8);
  if (fused.keys_out != nullptr && lane == 0) fused.keys_out[p] = winner;
  if (fused.records == nullptr) return;
  const int cw = static_cast<int>(static_cast<int64_t>(best_lo) - a.index_offset);
  const int block = cw / kWave;
  const float* trace = fused.trace + (static_cast<size_t>(p) * blocks + block) * fused.trace_pitch;
  const int rec_floats = 4 + 2 * n + 3 * (n + 1);
  float* __restrict__ rec = fused.records + static_cast<size_t>(p) * rec_floats;
  // Every entry of the record but two is ONE load from an address that depends on the entry alone: all of a pass are
  // requested (same instruction for trace, control matrix and start state: an agent-scope load), then all stored.
  constexpr int kSlots = 8;
  for (int e0 = 0; e0 < rec_floats; e0 += kSlots * kWave) {
    float value[kSlots];
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      if (e0 + q * kWave < rec_floats) {   // (wave-uniform)
        const int e = min(e0 + q * kWave + lane, rec_floats - 1);
        const int u = e - 4;
        const float* src = trace + (e - (4 + 2 * n + 3));                                        // states
        src = (e < 4 + 2 * n + 3) ? x0 + (e - (4 + 2 * n)) : src;                                // start state
        const float* from_u = (LAYOUT == 1) ? a.U + ((static_cast<size_t>(p) * n + (u >> 1)) * 2 + (u & 1)) * a.N + cw
                                            : a.U + (static_cast<size_t>(p) * a.N + cw) * 2 * n + u;
        src = (e < 4 + 2 * n) ? from_u : src;                                                    // controls
        src = (e < 4) ? trace + 3 * n + (1 - min(e, 1)) : src;                                   // cost, violation
        value[q] = observe(src);
      }
    }
#pragma unroll
    for (int q = 0; q < kSlots; ++q) {
      const int e = e0 + q * kWave + lane;
      if (e0 + q * kWave < rec_floats) {
        const float out = (e == 2) ? static_cast<float>(total_feas) : (e == 3) ? 1.0f : value[q];
        if (e < rec_floats) rec[e] = out;
      }
    }
  }
  ACMPC_STAMP(9);
}

// ---- softmin-weighted mean -------------------------------------------------------------------------------
constexpr int kSoftChunk = 1024;  // candidates per workgroup
constexpr int kSoftBlock = 256;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

// partial[p][chunk][0] = sum of weights; [1 .. 2n] = weighted sums; [2n+1 .. 4n] = unweighted sums.
template <int LAYOUT>
__global__ void __launch_bounds__(kSoftBlock) softmin_partial_kernel(const SoftminArgs a) {
  __shared__ float s_w[kSoftChunk];
  __shared__ double s_red[kSoftBlock / kWave];
  __shared__ double s_acc[2][kSoftBlock];
  const int p = blockIdx.y;
  const int chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int n2 = 2 * a.n;
  const int base = chunk * kSoftChunk;
  const int count = min(kSoftChunk, a.N - base);
  const float cmin = key_cost(a.keys[p]);
  const float* __restrict__ costs = a.costs + static_cast<size_t>(p) * a.N + base;
  double* __restrict__ out = a.partial + (static_cast<size_t>(p) * a.chunks + chunk) * (2 * n2 + 1);

  double wsum = 0.0;
  for (int c = tid; c < count; c += kSoftBlock) {
    const float cost = costs[c];
    const bool finite = (__float_as_uint(cost) & 0x7f800000u) != 0x7f800000u;
    const float wt = finite ? expf(-(cost - cmin) / a.lambda) : 0.0f;
    s_w[c] = wt;
    wsum += static_cast<double>(wt);
  }
  wsum = wave_sum_f64(wsum);
  if ((tid & (kWave - 1)) == 0) s_red[tid / kWave] = wsum;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int q = 0; q < kSoftBlock / kWave; ++q) t += s_red[q];
    out[0] = t;
  }

  if constexpr (LAYOUT == 0) {
    // rows of 2n floats: thread group g owns rows c = g (mod G); each thread a fixed entry e of the row
    const int G = kSoftBlock / n2 > 0 ? kSoftBlock / n2 : 1;
    for (int e0 = 0; e0 < n2; e0 += kSoftBlock) {  // n2 > 256 only for n > 128
      const int g = tid / n2;
      const int e = e0 + (tid % n2);
      double acc = 0.0, plain = 0.0;
      if (g < G && e < n2) {
        const float* __restrict__ U = a.U + (static_cast<size_t>(p) * a.N + base) * n2 + e;
        for (int c = g; c < count; c += G) {
          const double u = static_cast<double>(U[static_cast<size_t>(c) * n2]);
          const double wt = static_cast<double>(s_w[c]);
          if (wt != 0.0) acc += wt * u;  // a zero-weight (non-finite cost) candidate is excluded, NaN controls too
          plain += u;
        }
      }
      s_acc[0][tid] = acc;
      s_acc[1][tid] = plain;
      __syncthreads();
      if (tid < n2 && e0 + tid < n2) {
        double t0 = 0.0, t1 = 0.0;
        for (int g2 = 0; g2 < G; ++g2) {
          t0 += s_acc[0][g2 * n2 + tid];
          t1 += s_acc[1][g2 * n2 + tid];
        }
        out[1 + e0 + tid] = t0;
        out[1 + n2 + e0 + tid] = t1;
      }
      __syncthreads();
    }
  } else {
    // U[p][i][comp][N]: each WAVE owns entries e = wave, wave + 4, ... of the 2n, its lanes stride the chunk's
    // candidates (coalesced) and a shuffle reduction finishes the entry - no workgroup barrier per entry (a
    // workgroup-wide reduction per entry made this kernel 98 dependent barriers long: 630 us at N = 16 384)
    const int wave = tid / kWave;
    const int lane = tid & (kWave - 1);
    for (int e = wave; e < n2; e += kSoftBlock / kWave) {
      const float* __restrict__ U = a.U + (static_cast<size_t>(p) * n2 + e) * a.N + base;
      double acc = 0.0, plain = 0.0;
      for (int c = lane; c < count; c += kWave) {
        const double u = static_cast<double>(U[c]);
        const double wt = static_cast<double>(s_w[c]);
        if (wt != 0.0) acc += wt * u;
        plain += u;
      }
      acc = wave_sum_f64(acc);
      plain = wave_sum_f64(plain);
      if (lane == 0) {
        out[1 + e] = acc;
        out[1 + n2 + e] = plain;
      }
    }
  }
}

// Sums the chunk partials in chunk order; sum(w u)/sum(w), uniform weights when sum(w) is not positive
// (the NaN fallback of localiser.py:575-578).
__global__ void __launch_bounds__(kSoftBlock) softmin_final_kernel(const SoftminArgs a) {
  const int p = blockIdx.x;
  const int n2 = 2 * a.n;
  const double* __restrict__ part = a.partial + static_cast<size_t>(p) * a.chunks * (2 * n2 + 1);
  double wsum = 0.0;
  for (int q = 0; q < a.chunks; ++q) wsum += part[static_cast<size_t>(q) * (2 * n2 + 1)];
  const bool usable = wsum > 0.0;
  for (int e = threadIdx.x; e < n2; e += kSoftBlock) {
    double acc = 0.0;
    const int col = usable ? 1 + e : 1 + n2 + e;
    for (int q = 0; q < a.chunks; ++q) acc += part[static_cast<size_t>(q) * (2 * n2 + 1) + col];
    a.mean[static_cast<size_t>(p) * n2 + e] =
        static_cast<float>(acc / (usable ? wsum : static_cast<double>(a.N)));
  }
  if (threadIdx.x == 0 && a.weight_sum != nullptr) a.weight_sum[p] = wsum;
}

template <int MODE, int LAYOUT, int CPT, int BLOCK, int PACK = (CPT >= 2 ? 2 : 1), int WAVES = 1>
hipError_t launch_rollout_t(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s, hipEvent_t e0,
                            hipEvent_t e1) {
  const dim3 grid(shape.blocks_per_problem, args.P);
  const size_t lds = 64 + (MODE == 1 ? (static_cast<size_t>(args.n) * (kCoefT + kKeyStride) + 3 +
                                        (args.nn_frames != nullptr ? verified_frame_floats(args.n) : 0)) * sizeof(float)
                                     : 0);
  if (e0 != nullptr && e1 != nullptr) {
    hipExtLaunchKernelGGL((rollout_kernel<MODE, LAYOUT, CPT, BLOCK, PACK, WAVES>), grid, dim3(BLOCK),
                          static_cast<std::uint32_t>(lds), s, e0, e1, 0, args);
  } else {
    hipLaunchKernelGGL((rollout_kernel<MODE, LAYOUT, CPT, BLOCK, PACK, WAVES>), grid, dim3(BLOCK), lds, s, args);
  }
  return hipGetLastError();
}

template <int MODE>
hipError_t launch_rollout_tile(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s, hipEvent_t e0,
                               hipEvent_t e1) {
  const dim3 grid(shape.blocks_per_problem, args.P);
  const size_t lds = tile_lds_bytes(MODE, args.n);
  if (e0 != nullptr && e1 != nullptr) {
    hipExtLaunchKernelGGL((rollout_tile_kernel<MODE>), grid, dim3(kWave), static_cast<std::uint32_t>(lds), s, e0, e1,
                          0, args);
  } else {
    hipLaunchKernelGGL((rollout_tile_kernel<MODE>), grid, dim3(kWave), lds, s, args);
  }
  return hipGetLastError();
}

// Candidate-major, mode S, horizons of at most NMAX steps: the tile is only PASSED THROUGH the LDS.  rollout_tile_kernel
// keeps its 8n * 64 bytes of LDS for the whole walk, which caps a CU at six waves on four SIMDs.  Here a wave loads its
// span into registers (16-byte pieces, every line once), and the WAVES waves of a workgroup take turns at ONE tile
// buffer: write the pieces, read the own row back (ds_read_b64, conflict-free for odd n) into 2n registers, hand the
// buffer on.  The walk then runs out of registers with no LDS instruction in it, at the occupancy the registers allow
// (four waves per SIMD at H = 50), while other waves of the CU are still loading.
template <int NMAX, int WAVES, bool LDS_TABLE>
__global__ void __launch_bounds__(WAVES * kWave) rollout_tile_rows_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_tile[];  // ONE [64][2n] tile, used by the waves in turn
  const int p = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) / kWave);
  const int tiles = (a.N + kWave - 1) / kWave;
  const int tile = blockIdx.x * WAVES + wave;
  const bool live = tile < tiles;  // (wave-uniform; a workgroup's spare waves still take their turns at the barrier)
  const int c0 = tile * kWave;
  const int rows = live ? min(kWave, a.N - c0) : 0;
  const int n = a.n;
  const int row_floats = 2 * n;
  const Weights w = a.w;
  const float* __restrict__ coef = a.coef + static_cast<size_t>(p) * n * kCoefS;
  const float* __restrict__ x0 = a.x0 + p * 3;

  // the span starts on a 16-byte boundary (the launcher checks 2 N n % 4 == 0 or P == 1)
  const float* __restrict__ src = a.U + (static_cast<size_t>(p) * a.N + c0) * row_floats;
  const f32x4* __restrict__ src4 = reinterpret_cast<const f32x4*>(src);
  const int total = rows * row_floats;
  const int quads = total >> 2;
  constexpr int kQuads = (2 * NMAX + 3) / 4;  // 16-byte pieces per lane of a [64][2 NMAX] tile
  f32x4 raw[kQuads];
#pragma unroll
  for (int k = 0; k < kQuads; ++k) {
    const int q = lane + k * kWave;
    if (q < quads) raw[k] = __builtin_nontemporal_load(src4 + q);
  }
  f32x2 rest = {0.0f, 0.0f};
  const bool has_rest = (total & 2) != 0 && lane == 0;  // rows * n odd: one (v, kappa) pair past the last full piece
  if (has_rest) rest = *reinterpret_cast<const f32x2*>(src + (quads << 2));

  // turns at the one buffer: wave t goes after t barriers and leaves WAVES - 1 - t behind it (every wave passes the
  // same WAVES - 1 barriers; the loads above are in flight while a wave waits for its turn)
  auto handover = []() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  // The workgroup's four tiles belong to ONE problem: its table ([n][12] floats) goes into LDS once, behind the tile,
  // and the walk reads its rows from there one step ahead (every lane the same address: a broadcast).  A scalar load
  // per step misses the scalar cache (the tables of 256 problems do not fit it) and a wave then waits longer than it
  // computes: 59 % of the wave-cycles of the scalar-load form are waits.
  float* const s_table = s_tile + ((kWave * row_floats + 3) & ~3);
  if constexpr (LDS_TABLE) {
    const f32x4* __restrict__ coef4 = reinterpret_cast<const f32x4*>(coef);
    f32x4* table4 = reinterpret_cast<f32x4*>(s_table);
    for (int q = threadIdx.x; q < 3 * n; q += WAVES * kWave) table4[q] = coef4[q];
    handover();
  }
  for (int t = 0; t < wave; ++t) handover();
  {
    f32x4* dst4 = reinterpret_cast<f32x4*>(s_tile);
#pragma unroll
    for (int k = 0; k < kQuads; ++k) {
      const int q = lane + k * kWave;
      if (q < quads) dst4[q] = raw[k];
    }
    if (has_rest) *reinterpret_cast<f32x2*>(s_tile + (quads << 2)) = rest;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  f32x2 u[NMAX];
  {
    const f32x2* row = reinterpret_cast<const f32x2*>(s_tile + lane * row_floats);
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
      if (i < n) u[i] = row[i];
  }
  for (int t = wave; t < WAVES - 1; ++t) handover();

  const bool active = lane < rows;
  float cost = __builtin_inff();
  bool feas = false;
  if (active) {
    StateS st{x0[0], x0[1], x0[2], 0.0f, 0.0f};
    if constexpr (LDS_TABLE) {
      constexpr int kRowUsed = 9;
      float row_now[kRowUsed], row_next[kRowUsed];
      auto fetch = [&](float (&dst)[kRowUsed], int i) {   // (rows past n: whatever the LDS holds there, never used)
#pragma unroll
        for (int j = 0; j < kRowUsed; ++j) dst[j] = s_table[i * kCoefS + j];
      };
      fetch(row_now, 0);
#pragma unroll
      for (int i = 0; i < NMAX; ++i) {
        if (i + 1 < NMAX) fetch(row_next, i + 1);
        if (i < n) step_spatial(st, row_now, u[i][0], u[i][1], w);
#pragma unroll
        for (int j = 0; j < kRowUsed; ++j) row_now[j] = row_next[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NMAX; ++i)
        if (i < n) step_spatial(st, coef + i * kCoefS, u[i][0], u[i][1], w);
    }
    cost = finish_spatial(st, w);
    feas = st.V == 0.0f;
    if (a.costs != nullptr) a.costs[static_cast<size_t>(p) * a.N + c0 + lane] = cost;
  }
  int64_t key = active ? pack_key(cost, static_cast<uint32_t>(a.index_offset + c0 + lane)) : kKeyMax;
  int nfeas = (active && feas) ? 1 : 0;
  key = wave_min_key(key);
  nfeas = wave_sum_int(nfeas);
  if (lane == 0 && live) {
    const size_t slot = static_cast<size_t>(p) * tiles + tile;
    a.partial_keys[slot] = key;
    a.partial_feas[slot] = nfeas;
  }
}

template <int NMAX, int WAVES, bool LDS_TABLE>
hipError_t launch_rollout_tile_rows(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s, hipEvent_t e0,
                                    hipEvent_t e1) {
  const dim3 grid((shape.blocks_per_problem + WAVES - 1) / WAVES, args.P);
  // the tile, then (LDS_TABLE) the problem's table with room for NMAX rows (the walk's look-ahead reads that far)
  const size_t lds = tile_lds_bytes(0, args.n) + (LDS_TABLE ? static_cast<size_t>(NMAX + 1) * kCoefS * sizeof(float) : 0);
  if (e0 != nullptr && e1 != nullptr) {
    hipExtLaunchKernelGGL((rollout_tile_rows_kernel<NMAX, WAVES, LDS_TABLE>), grid, dim3(WAVES * kWave),
                          static_cast<std::uint32_t>(lds), s, e0, e1, 0, args);
  } else {
    hipLaunchKernelGGL((rollout_tile_rows_kernel<NMAX, WAVES, LDS_TABLE>), grid, dim3(WAVES * kWave), lds, s, args);
  }
  return hipGetLastError();
}

#ifndef ACMPC_TEMPORAL_TU
template <int MODE, int LAYOUT>
hipError_t launch_rollout_ml(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s, hipEvent_t e0,
                             hipEvent_t e1) {
  if constexpr (MODE == 1 && LAYOUT == 1) {
    // plain float32 arithmetic, one state per candidate: built in its own translation unit (acmpc_kernels_temporal.hip)
    if (shape.pack == 1) return launch_rollout_temporal_plain(shape, args, s, e0, e1);
  }
  if constexpr (LAYOUT == 0) {
    if constexpr (MODE == 0) {
      // (the rows kernel moves 16-byte pieces: a control matrix that does not start on a 16-byte boundary - a view
      // into a caller's buffer - takes the other kernel)
      if (shape.tile && shape.tile_waves == 4 && (reinterpret_cast<uintptr_t>(args.U) & 15u) == 0)
        return launch_rollout_tile_rows_plain(shape, args, s, e0, e1);
    }
    if (shape.tile) return launch_rollout_tile<MODE>(shape, args, s, e0, e1);
  }
  if (shape.block == 64 && shape.cpt == 1) return launch_rollout_t<MODE, LAYOUT, 1, 64>(shape, args, s, e0, e1);
  if (shape.block == 256 && shape.cpt == 1) return launch_rollout_t<MODE, LAYOUT, 1, 256>(shape, args, s, e0, e1);
  if constexpr (LAYOUT == 1) {
    if (shape.block == 256 && shape.cpt == 2) return launch_rollout_t<MODE, LAYOUT, 2, 256>(shape, args, s, e0, e1);
    if (shape.block == 256 && shape.cpt == 4) return launch_rollout_t<MODE, LAYOUT, 4, 256>(shape, args, s, e0, e1);
  }
  return hipErrorInvalidConfiguration;
}
#endif  // ACMPC_TEMPORAL_TU

}  // namespace

#ifdef ACMPC_TEMPORAL_TU
// Mode T on the step-major layout with one arithmetic state per candidate (plain v_*_f32 instructions).  The kernel is
// bound by instruction issue and by the latency of its two dependent LDS gathers per step, not by HBM: measured on
// MI355X (1 M candidates per launch) the compiler's SLP re-packing of neighbouring candidates into v_pk_* pairs costs
// 15 % at the 8-waypoint window (183 -> 155 us), because a packed instruction issues at half the rate of a plain one
// and the packing adds moves - so this translation unit is compiled with -fno-slp-vectorize (ac-mpc_amd/acmpc_amd/_build.py).
// (the candidate-major rows kernel gains the same way: its walk is one candidate per lane, and the SLP vectoriser's
// ten packed instructions + five moves per step cost more than the twenty plain ones they replace)
hipError_t launch_rollout_tile_rows_plain(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s,
                                          hipEvent_t e0, hipEvent_t e1) {
  // Table rows from LDS (one copy per workgroup) or by scalar loads - measured, 4 096 candidates per problem, LDS / scalar:
  //   H = 50:  256 problems 92 / 92 us, 1 024: 335 / 389, 4 096: 1 258 / 1 457   (more tables in flight, more scalar misses)
  //   H = 30:  256: 51.5 / 47.8, 1 024: 212 / 216;   H = 20, 2 048 problems: 297 / 280
  //   H = 65:  256: 183 / 173, 1 024: 654 / 685;     H = 80: 256: 221 / 220, 1 024: 799 / 1 094
  const bool lds = shape.tile_table != 0 ? shape.tile_table == 1 : (args.n > 32 && (args.n <= 50 || args.P >= 512));   // (A/B: LaunchOptions::tile_table)
  if (lds) {
    if (args.n <= 32) return launch_rollout_tile_rows<32, 4, true>(shape, args, s, e0, e1);
    if (args.n <= 50) return launch_rollout_tile_rows<50, 4, true>(shape, args, s, e0, e1);
    if (args.n <= 64) return launch_rollout_tile_rows<64, 4, true>(shape, args, s, e0, e1);
    return launch_rollout_tile_rows<kTileRowsMaxSteps, 4, true>(shape, args, s, e0, e1);
  }
  if (args.n <= 32) return launch_rollout_tile_rows<32, 4, false>(shape, args, s, e0, e1);
  if (args.n <= 50) return launch_rollout_tile_rows<50, 4, false>(shape, args, s, e0, e1);
  if (args.n <= 64) return launch_rollout_tile_rows<64, 4, false>(shape, args, s, e0, e1);
  return launch_rollout_tile_rows<kTileRowsMaxSteps, 4, false>(shape, args, s, e0, e1);
}

#ifdef ACMPC_T_STAMPS
}  // namespace acmpc
extern "C" int acmpc_debug_t_stamps(unsigned long long* out, int waves) {
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(acmpc::g_t_stamps), static_cast<size_t>(waves) * 6 * sizeof(unsigned long long)));
}
namespace acmpc {
#endif
hipError_t launch_rollout_temporal_plain(const LaunchShape& shape, const RolloutArgs& args, hipStream_t s,
                                         hipEvent_t e0, hipEvent_t e1) {
  if (shape.block == 64 && shape.cpt == 1) return launch_rollout_t<1, 1, 1, 64, 1>(shape, args, s, e0, e1);
  if (shape.block == 256 && shape.cpt == 1) return launch_rollout_t<1, 1, 1, 256, 1>(shape, args, s, e0, e1);
  if (shape.block == 256 && shape.cpt == 2) {
    // two candidates per lane, the allocation capped for eight waves per SIMD (62 VGPRs either way since round 4's key
    // table).  Round 5: ONE instantiation for every search.  The uncapped one the 8-waypoint window used to take
    // (next_free_sgpr 74 against 72, otherwise the same resources) was dealt badly by the dispatcher in every launch
    // looked at: of 2 048 workgroups - eight per compute unit, all of which fit - 12 to 60 were held back until a first
    // workgroup had finished, 65 us into a 130 us launch, beside compute units that ran seven all along; this one starts
    // all 8 192 waves within 1.6 us, eight per SIMD (tools/modeT_stamps.py).
    RolloutArgs one = args;
    one.even_progress = static_cast<long long>(args.P) * shape.blocks_per_problem * (shape.block / kWave) <= 8 * 1024 ? 1 : 0;
    return launch_rollout_t<1, 1, 2, 256, 1, 8>(shape, one, s, e0, e1);
  }
  if (shape.block == 256 && shape.cpt == 4) return launch_rollout_t<1, 1, 4, 256, 1>(shape, args, s, e0, e1);
  return hipErrorInvalidConfiguration;
}
#else

// hipGetLastError() returns (and clears) the last error of ANY earlier runtime call of the thread - a failed
// allocation of this or another library minutes ago included.  Launch status is read with it, so clear it first.
static inline void clear_stale_error() { (void)hipGetLastError(); }

int max_blocks_per_problem(int N) { return (N + kWave - 1) / kWave; }

size_t tile_lds_bytes(int mode, int n) {
  const size_t tile = (static_cast<size_t>(kWave) * 2 * n + 3) & ~static_cast<size_t>(3);
  return (tile + (mode == 1 ? static_cast<size_t>(n) * (kCoefT + kKeyStride) : 0)) * sizeof(float);
}

LaunchShape choose_shape(int P, int N, int layout, int mode, int n, const LaunchOptions& opt) {
  // Fill 256 CUs first (small batches: 64-thread workgroups, one candidate per lane), then widen the
  // per-lane work so that each wave load moves 16 B per lane (large step-major batches).
  LaunchShape s;
  s.tile = false;
  s.tile_waves = 0;
  s.pack = (mode == 1) ? 1 : 2;  // mode T: plain float32 states (see launch_rollout_temporal_plain)
  if (opt.temporal_pack != 0) s.pack = opt.temporal_pack;
  s.tile_table = opt.tile_table;
  const long long total = static_cast<long long>(P) * N;
  if (layout == 0 && tile_lds_bytes(mode, n) <= 64 * 1024 && !opt.no_tile) {
    // candidate-major: one wave per workgroup stages its 64 rows in LDS (rollout_tile_kernel)
    s.tile = true;
    s.block = kWave;
    s.cpt = 1;
    s.blocks_per_problem = (N + kWave - 1) / kWave;
    // mode S up to kTileRowsMaxSteps steps: rows in registers, the LDS tile shared by the waves of a workgroup in turn
    // (needs every problem's span on a 16-byte boundary)
    // - from 2 048 tiles up: below that the four-wave workgroups leave CUs idle (16 x 320 x 49: 19 us against 12)
    if (mode == 0 && n <= kTileRowsMaxSteps && (P == 1 || (2LL * N * n) % 4 == 0) &&
        static_cast<long long>(P) * s.blocks_per_problem >= 2048) {
      s.tile_waves = 4;
      if (opt.tile_rows >= 0) s.tile_waves = (opt.tile_rows == 4) ? 4 : 0;
    }
    return s;
  }
  // tuning override for experiments: ACMPC_SHAPE="<block>,<cpt>"
  int fb = opt.shape_block, fc = opt.shape_cpt;
  if (!((fb == 64 && fc == 1) || (fb == 256 && (fc == 1 || (layout == 1 && (fc == 2 || fc == 4) && N % fc == 0)))))
    fb = fc = 0;
  if (fb != 0) {
    s.block = fb;
    s.cpt = fc;
  } else if (total <= 256LL * 64 * 8) {
    s.block = 64;
    s.cpt = 1;
  } else if (mode == 1 && layout == 1 && N % 2 == 0) {
    // mode T waits on LDS gathers: two candidates per lane keep twice the waves in flight that four would at the same
    // batch size (1 M candidates: 155 us against 175 us at the 8-waypoint window)
    s.block = 256;
    s.cpt = 2;
  } else if (layout == 1 && N % 4 == 0 && total >= 256LL * 4096) {
    // from 1 M candidates up: four candidates per lane as two packed pairs (v_pk_* arithmetic, 16-byte loads).
    // Same-box A/B on 256 x 4 096 x 49 / 1 024 x 4 096 x 49: one per lane 72 / 290 us, two 85 / 285 us, four 68 / 283 us.
    s.block = 256;
    s.cpt = 4;
  } else {
    s.block = 256;
    s.cpt = 1;
  }
  const int per_block = s.block * s.cpt;
  s.blocks_per_problem = (N + per_block - 1) / per_block;
  return s;
}

hipError_t launch_rollout(int mode, int layout, const LaunchShape& shape, const RolloutArgs& args, hipStream_t s,
                          hipEvent_t e0, hipEvent_t e1) {
  clear_stale_error();
  if (mode == 0 && layout == 0) return launch_rollout_ml<0, 0>(shape, args, s, e0, e1);
  if (mode == 0 && layout == 1) return launch_rollout_ml<0, 1>(shape, args, s, e0, e1);
  if (mode == 1 && layout == 0) return launch_rollout_ml<1, 0>(shape, args, s, e0, e1);
  if (mode == 1 && layout == 1) return launch_rollout_ml<1, 1>(shape, args, s, e0, e1);
  return hipErrorInvalidValue;
}

bool tailed_rollout_fits(int mode, int layout, const LaunchShape& shape, int n) {
  const size_t rec_floats = static_cast<size_t>(4 + 2 * n + 3 * (n + 1));
  return mode == 0 && layout == 1 && !shape.tile && shape.block == 256 && (shape.cpt == 1 || shape.cpt == 2 || shape.cpt == 4) &&
         64 + rec_floats * sizeof(float) <= 64 * 1024;
}

hipError_t launch_rollout_tailed(int layout, const LaunchShape& shape, const RolloutArgs& args, const FinalizeArgs& fin,
                                 int* tickets, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  clear_stale_error();
  if (!tailed_rollout_fits(0, layout, shape, args.n) || tickets == nullptr) return hipErrorInvalidValue;
  const dim3 grid(shape.blocks_per_problem, args.P);
  const size_t lds = 64 + static_cast<size_t>(4 + 2 * args.n + 3 * (args.n + 1)) * sizeof(float);
  auto go = [&](auto kernel) -> hipError_t {
    if (e0 != nullptr && e1 != nullptr) {
      hipExtLaunchKernelGGL(kernel, grid, dim3(256), static_cast<std::uint32_t>(lds), s, e0, e1, 0, args, fin, tickets);
    } else {
      hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, args, fin, tickets);
    }
    return hipGetLastError();
  };
  if (shape.cpt == 1) return go(rollout_tailed_kernel<0, 1, 1, 256, 1>);
  if (shape.cpt == 2) return go(rollout_tailed_kernel<0, 1, 2, 256, 2>);
  return go(rollout_tailed_kernel<0, 1, 4, 256, 2>);
}

// rollout of one batch + finalize of the batch before it in one launch (rollout_chained_kernel): mode S, step-major
// rollout on the 256-thread shapes; a pending finalize of the many-problem kind whose winners are re-drawn or read from a
// step-major matrix, and whose LDS image leaves the rollout its eight workgroups per CU
bool chained_rollout_fits(int mode, int layout, const LaunchShape& shape, int P, const FinalizeArgs& fin, int fin_layout) {
  const int fin_groups = (fin.P + kFinalizeWaves - 1) / kFinalizeWaves;
  const int rows = (fin_groups + std::max(shape.blocks_per_problem, 1) - 1) / std::max(shape.blocks_per_problem, 1);
  return rows <= P && static_cast<long long>(P) + rows <= 65535 && mode == 0 && layout == 1 && !shape.tile && shape.block == 256 && (shape.cpt == 1 || shape.cpt == 2 || shape.cpt == 4) &&
         !fin.controls_only && fin.records != nullptr && fin.keys_in == nullptr && fin.P >= 1 && (fin.regenerate || fin_layout == 1) &&
         kFinalizeWaves * group_finalize_floats(fin.n, 1) * sizeof(float) <= 20 * 1024;
}

hipError_t launch_rollout_chained(int layout, const LaunchShape& shape, const RolloutArgs& args, const FinalizeArgs& fin,
                                  int fin_layout, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
  clear_stale_error();
  if (!chained_rollout_fits(0, layout, shape, args.P, fin, fin_layout)) return hipErrorInvalidValue;
  const int groups = (fin.P + kFinalizeWaves - 1) / kFinalizeWaves;
  const int fin_rows = (groups + shape.blocks_per_problem - 1) / shape.blocks_per_problem;
  const dim3 grid(shape.blocks_per_problem, args.P + fin_rows);
  const size_t lds = std::max<size_t>(64, kFinalizeWaves * group_finalize_floats(fin.n, 1) * sizeof(float));
  auto go = [&](auto kernel) -> hipError_t {
    if (e0 != nullptr && e1 != nullptr) {
      hipExtLaunchKernelGGL(kernel, grid, dim3(256), static_cast<std::uint32_t>(lds), s, e0, e1, 0, args, fin);
    } else {
      hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, args, fin);
    }
    return hipGetLastError();
  };
  if (shape.cpt == 1) return go(rollout_chained_kernel<1, 1, 256, 1>);
  if (shape.cpt == 2) return go(rollout_chained_kernel<1, 2, 256, 2>);
  return go(rollout_chained_kernel<1, 4, 256, 2>);
}

hipError_t launch_finalize(int mode, int layout, const FinalizeArgs& args, hipStream_t s, const LaunchOptions& opt) {
  clear_stale_error();
  if (mode == 0 && !args.controls_only && args.P >= kGroupFinalizeProblems && !opt.no_group_finalize &&
      (layout == 0 || layout == 1) && group_finalize_floats(args.n) * sizeof(float) <= 64 * 1024) {
    if (opt.finalize_waves && kFinalizeWaves * group_finalize_floats(args.n, 1) * sizeof(float) <= 64 * 1024) {
      const dim3 waves_grid((args.P + kFinalizeWaves - 1) / kFinalizeWaves);
      const size_t waves_lds = kFinalizeWaves * group_finalize_floats(args.n, 1) * sizeof(float);
      if (layout == 0) {
        hipLaunchKernelGGL((finalize_waves_kernel<0>), waves_grid, dim3(kFinalizeWaves * kWave), waves_lds, s, args);
      } else {
        hipLaunchKernelGGL((finalize_waves_kernel<1>), waves_grid, dim3(kFinalizeWaves * kWave), waves_lds, s, args);
      }
      return hipGetLastError();
    }
    // many problems: sixteen lanes per problem, four problems per wavefront (finalize_groups_kernel)
    const dim3 groups_grid((args.P + kGroupProblems - 1) / kGroupProblems);
    const size_t groups_lds = group_finalize_floats(args.n) * sizeof(float);
    if (layout == 0) {
      hipLaunchKernelGGL((finalize_groups_kernel<0>), groups_grid, dim3(kWave), groups_lds, s, args);
    } else {
      hipLaunchKernelGGL((finalize_groups_kernel<1>), groups_grid, dim3(kWave), groups_lds, s, args);
    }
    return hipGetLastError();
  }
  const dim3 grid(args.P), block(kWave);
  // record image, then (mode T) the waypoint table
  const size_t rec_floats = static_cast<size_t>(4 + 2 * args.n + 3 * (args.n + 1));
  const size_t lds = (((rec_floats + 3) & ~static_cast<size_t>(3)) + (mode == 1 ? args.n * (kCoefT + kKeyStride) : 0)) * sizeof(float);
  if (mode == 0 && layout == 0) {
    hipLaunchKernelGGL((finalize_kernel<0, 0>), grid, block, lds, s, args);
  } else if (mode == 0 && layout == 1) {
    hipLaunchKernelGGL((finalize_kernel<0, 1>), grid, block, lds, s, args);
  } else if (mode == 1 && layout == 0) {
    hipLaunchKernelGGL((finalize_kernel<1, 0>), grid, block, lds, s, args);
  } else if (mode == 1 && layout == 1) {
    hipLaunchKernelGGL((finalize_kernel<1, 1>), grid, block, lds, s, args);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_sample(int layout, const SampleArgs& args, hipStream_t s) {
  clear_stale_error();
  const dim3 grid((args.N + 255) / 256, args.P);
  const size_t sample_lds = (((static_cast<size_t>(args.n) + 3) & ~static_cast<size_t>(3)) + 2 * args.n) * sizeof(float);
  if (layout == 0) {
    hipLaunchKernelGGL((sample_kernel<0>), grid, dim3(256), sample_lds, s, args);
  } else if (layout == 1) {
    hipLaunchKernelGGL((sample_kernel<1>), grid, dim3(256), sample_lds, s, args);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

namespace {
size_t sampled_rollout_floats(int mode, int n) {
  return (mode == 1) ? ((static_cast<size_t>(n) * (kCoefT + kKeyStride) + 3) & ~static_cast<size_t>(3)) : 0;
}
// uniform operands of the steps staged in LDS: [n][12] table rows (mode S), centre, reference, candidate 2's controls, knot weights (+ padding
// for the read one step past the end)
size_t sampled_uniform_floats(int mode, int n) {
  return static_cast<size_t>(n) * ((mode == 0 ? kCoefS : 0) + 7) + 4;
}
size_t sampled_finalize_floats(int mode, int n) {
  const size_t rec_floats = static_cast<size_t>(4 + 2 * n + 3 * (n + 1));
  return ((rec_floats + 3) & ~static_cast<size_t>(3)) + (mode == 1 ? static_cast<size_t>(n) * (kCoefT + kKeyStride) : 0);
}
}  // namespace

bool fused_finalize_fits(int mode, int n) {
  return (sampled_rollout_floats(mode, n) + sampled_finalize_floats(mode, n) + sampled_uniform_floats(mode, n)) *
             sizeof(float) <= 64 * 1024;
}

int trace_floats(int n) { return 5 * n + 2; }

// whether the frames of the verified search fit beside the three-wave mode T round's tables, trace, operands and indices
// (launch_rollout_sampled drops them otherwise and the search wave scans every waypoint): they do up to n = 106
bool trio_frames_fit(int n) {
  if (n < kVerifiedWindow) return false;
  const size_t tables = ((static_cast<size_t>(n) * (kCoefT + kKeyStride) + 3) & ~static_cast<size_t>(3)) +
                        static_cast<size_t>(verified_frame_floats(n));
  const size_t trace = static_cast<size_t>(trace_floats(n)) * kWave;
  const size_t uniform = (static_cast<size_t>(n) * 7 + 4 + 3) & ~static_cast<size_t>(3);
  const size_t index = (static_cast<size_t>(n) * kWave / 2 + 3) & ~static_cast<size_t>(3);
  return (tables + trace + uniform + index) * sizeof(float) <= 160u * 1024u;
}

// The traced form keeps [5n + 2][64] floats in LDS per workgroup (63 kB at H = 50): up to the CU's 160 kB.
bool traced_finalize_fits(int mode, int n) {
  return (sampled_rollout_floats(mode, n) + static_cast<size_t>(trace_floats(n)) * kWave +
          sampled_uniform_floats(mode, n)) * sizeof(float) <= 160 * 1024;
}

// More dynamic LDS than a kernel gets by default (64 kB): raise the kernel's limit, once per kernel and device.
static hipError_t raise_lds_limit(const void* kernel, int which, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  static bool raised[13][64] = {};
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  if (device < 0 || device >= 64) return hipErrorInvalidDevice;
  if (!raised[which][device]) {
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    raised[which][device] = true;
  }
  return hipSuccess;
}

hipError_t launch_rollout_sampled(int mode, const RolloutArgs& rollout, const SampleArgs& sample,
                                  const FusedFinalize& fused, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                                  const LaunchOptions& opt) {
  clear_stale_error();
  const int n = rollout.n;
  const dim3 grid((rollout.N + kWave - 1) / kWave, rollout.P);
  const size_t rollout_floats = sampled_rollout_floats(mode, n);
  const bool traced = fused.trace != nullptr;
  const size_t finalize_floats = traced                     ? static_cast<size_t>(trace_floats(n)) * kWave
                                 : fused.tickets != nullptr ? sampled_finalize_floats(mode, n)
                                                            : 0;
  const size_t uniform_floats = sampled_uniform_floats(mode, n);
  const size_t lds = (rollout_floats + finalize_floats + uniform_floats) * sizeof(float);
  if (lds > 160u * 1024u) return hipErrorInvalidValue;  // callers check *_fits() first
  const int uniform_offset = static_cast<int>(rollout_floats + finalize_floats);
  if (traced && fused.trace_pitch < trace_floats(n)) return hipErrorInvalidValue;
  const int offset = static_cast<int>(rollout_floats);
  if (mode != 0 && mode != 1) return hipErrorInvalidValue;
  {
    const hipError_t e = raise_lds_limit(mode == 0 ? reinterpret_cast<const void*>(&rollout_sampled_kernel<0>)
                                                   : reinterpret_cast<const void*>(&rollout_sampled_kernel<1>),
                                         mode, lds);
    if (e != hipSuccess) return e;
  }
  if (mode == 1 && traced && !opt.no_trio_rounds) {
    // three waves per workgroup: tables | trace | uniform operands (centre, reference, weights) | nearest indices
    const size_t plain_tables = (static_cast<size_t>(n) * (kCoefT + kKeyStride) + 3) & ~static_cast<size_t>(3);
    const size_t trace = static_cast<size_t>(trace_floats(n)) * kWave;
    const size_t uniform = (static_cast<size_t>(n) * 7 + 4 + 3) & ~static_cast<size_t>(3);
    const size_t index = (static_cast<size_t>(n) * kWave / 2 + 3) & ~static_cast<size_t>(3);   // 16-bit entries
    // the frames of the verified search ride along when they fit beside the rest (they do up to n = 106); a longer
    // horizon keeps the three waves and scans every waypoint, as it did before there were frames
    RolloutArgs rollout_trio = rollout;
    size_t tables = plain_tables + (rollout.nn_frames != nullptr ? static_cast<size_t>(verified_frame_floats(n)) : 0);
    if ((tables + trace + uniform + index) * sizeof(float) > 160u * 1024u) {
      rollout_trio.nn_frames = nullptr;
      tables = plain_tables;
    }
    const size_t trio_lds = (tables + trace + uniform + index) * sizeof(float);
    if (trio_lds <= 160u * 1024u) {
      hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&rollout_sampled_trio_kernel), 11, trio_lds);
      if (e != hipSuccess) return e;
      if (e0 != nullptr && e1 != nullptr) {
        hipExtLaunchKernelGGL(rollout_sampled_trio_kernel, grid, dim3(3 * kWave), static_cast<std::uint32_t>(trio_lds), s, e0,
                              e1, 0, rollout_trio, sample, fused, static_cast<int>(tables), static_cast<int>(tables + trace),
                              static_cast<int>(tables + trace + uniform));
      } else {
        hipLaunchKernelGGL(rollout_sampled_trio_kernel, grid, dim3(3 * kWave), trio_lds, s, rollout_trio, sample, fused,
                           static_cast<int>(tables), static_cast<int>(tables + trace),
                           static_cast<int>(tables + trace + uniform));
      }
      return hipGetLastError();
    }
  }
  if (mode == 0 && traced && !opt.no_quad_rounds && !opt.no_pair_rounds) {
    // four waves per workgroup: trace | uniform operands (table rows, centre, reference, weights) | normals | J
    const size_t trace = static_cast<size_t>(trace_floats(n)) * kWave;
    const size_t uniform = (uniform_floats + 3) & ~static_cast<size_t>(3);
    const size_t quad_lds = (trace + uniform + static_cast<size_t>(2 * kKnots + 1) * kWave) * sizeof(float);
    if (quad_lds <= 160u * 1024u) {
      hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&rollout_sampled_quad_kernel), 12, quad_lds);
      if (e != hipSuccess) return e;
      if (e0 != nullptr && e1 != nullptr) {
        hipExtLaunchKernelGGL(rollout_sampled_quad_kernel, grid, dim3(kQuadWaves * kWave), static_cast<std::uint32_t>(quad_lds), s, e0,
                              e1, 0, rollout, sample, fused, static_cast<int>(trace), static_cast<int>(trace + uniform));
      } else {
        hipLaunchKernelGGL(rollout_sampled_quad_kernel, grid, dim3(kQuadWaves * kWave), quad_lds, s, rollout, sample, fused,
                           static_cast<int>(trace), static_cast<int>(trace + uniform));
      }
      return hipGetLastError();
    }
  }
  if (mode == 0 && traced && !opt.no_pair_rounds) {
    // two waves per workgroup: trace | uniform operands | exchange buffers (no mode T tables, no record image)
    const size_t trace = static_cast<size_t>(trace_floats(n)) * kWave;
    const size_t exchange = static_cast<size_t>(2) * kPairChunk * kPairValues * kWave;
    const size_t pair_lds = (trace + uniform_floats + exchange) * sizeof(float);
    if (pair_lds <= 160u * 1024u) {
      hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&rollout_sampled_pair_kernel), 2, pair_lds);
      if (e != hipSuccess) return e;
      if (e0 != nullptr && e1 != nullptr) {
        hipExtLaunchKernelGGL(rollout_sampled_pair_kernel, grid, dim3(2 * kWave), static_cast<std::uint32_t>(pair_lds), s,
                              e0, e1, 0, rollout, sample, fused, static_cast<int>(trace),
                              static_cast<int>(trace + uniform_floats));
      } else {
        hipLaunchKernelGGL(rollout_sampled_pair_kernel, grid, dim3(2 * kWave), pair_lds, s, rollout, sample, fused,
                           static_cast<int>(trace), static_cast<int>(trace + uniform_floats));
      }
      return hipGetLastError();
    }
  }
  const bool timed = e0 != nullptr && e1 != nullptr;
  const std::uint32_t lds32 = static_cast<std::uint32_t>(lds);
  if (mode == 0 && timed) {
    hipExtLaunchKernelGGL((rollout_sampled_kernel<0>), grid, dim3(kWave), lds32, s, e0, e1, 0, rollout, sample, fused, offset, uniform_offset);
  } else if (mode == 0) {
    hipLaunchKernelGGL((rollout_sampled_kernel<0>), grid, dim3(kWave), lds, s, rollout, sample, fused, offset, uniform_offset);
  } else if (timed) {
    hipExtLaunchKernelGGL((rollout_sampled_kernel<1>), grid, dim3(kWave), lds32, s, e0, e1, 0, rollout, sample, fused, offset, uniform_offset);
  } else {
    hipLaunchKernelGGL((rollout_sampled_kernel<1>), grid, dim3(kWave), lds, s, rollout, sample, fused, offset, uniform_offset);
  }
  return hipGetLastError();
}

// acmpc_solve_device's one-launch form (mode S): see rollout_solo_kernel.
constexpr int kSoloRegisterSteps = 49;   // the horizon whose states stay in registers (H = 50)

// Registers or LDS for the states, measured (device-resident solve, p50 of 300, three engines each; registers / LDS):
//   step-major   4 096 x 49: 11.6 / 12.4 us    16 384 x 49: 12.5 / 12.6    65 536 x 49: 17.2 / 15.9
//   cand.-major  4 096 x 49: 12.7 / 13.9                                    65 536 x 49: 18.7 / (two launches: 26.7)
// so: registers up to 512 workgroups (no SIMD holds more than one wave), and in the candidate-major layout always -
// there the 38 kB of states beside the 25 kB control tile would keep a launch of more than 512 workgroups from being
// resident at once.
static bool solo_in_registers(long long blocks, int n, int layout, const LaunchOptions& opt) {
  if (n != kSoloRegisterSteps) return false;
  if (opt.solo_registers >= 0) return opt.solo_registers == 1;   // (A/B switch)
  return blocks <= 512 || layout == 0;
}

static size_t solo_lds_bytes(long long blocks, int layout, int n, const LaunchOptions& opt) {
  const size_t states = solo_in_registers(blocks, n, layout, opt) ? static_cast<size_t>((3 * n + 3) & ~3)
                                                             : static_cast<size_t>((3 * n) | 1) * kWave;
  return (states + kWave + (layout == 0 ? static_cast<size_t>(kWave) * 2 * n : 0)) * sizeof(float);
}

bool solo_fits(int P, int N, int n, int layout, const LaunchOptions& opt) {
  // every workgroup of the launch resident at once (256 CUs x 160 kB of LDS; eight two-wave workgroups per CU): a second
  // generation of workgroups would cost more than the second launch does
  const long long blocks = static_cast<long long>(P) * ((N + kWave - 1) / kWave);
  const size_t lds = solo_lds_bytes(blocks, layout, n, opt);
  return blocks <= kSoloBlocks && lds <= 160u * 1024u &&
         blocks <= 256LL * std::min<long long>(8, static_cast<long long>((160u * 1024u) / lds));
}

int solo_trace_floats(int n) { return 3 * n + 2; }

hipError_t launch_rollout_solo(int layout, const RolloutArgs& args, const FusedFinalize& fused_in, hipStream_t s,
                               hipEvent_t e0, hipEvent_t e1, const LaunchOptions& opt) {
  clear_stale_error();
  const int blocks = (args.N + kWave - 1) / kWave;
  if (!solo_fits(args.P, args.N, args.n, layout, opt) || fused_in.tickets == nullptr) return hipErrorInvalidValue;
  if (fused_in.records != nullptr && (fused_in.trace == nullptr || fused_in.trace_pitch < solo_trace_floats(args.n)))
    return hipErrorInvalidValue;
  FusedFinalize fused = fused_in;
  // ticket groups: 8 for launches of up to 256 workgroups, 32 above (a device-scope atomic on one address is ~13 ns)
  fused.ticket_groups = (args.P * blocks > 256 || blocks > 256) ? kTicketGroupsMax : kTicketGroups;
  // two waves per workgroup (see the kernel); ACMPC_SOLO_SPLIT=0 keeps one, for the tests' three-way comparison
  const bool split = opt.solo_split < 0 || opt.solo_split == 1;
  const long long all_blocks = static_cast<long long>(args.P) * blocks;
  const size_t lds = solo_lds_bytes(all_blocks, layout, args.n, opt);
  const dim3 grid(blocks, args.P);
  auto go = [&](auto kernel, int which, int threads) -> hipError_t {
    const hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kernel), which, lds);
    if (e != hipSuccess) return e;
    if (e0 != nullptr && e1 != nullptr) {
      hipExtLaunchKernelGGL(kernel, grid, dim3(threads), static_cast<std::uint32_t>(lds), s, e0, e1, 0, args, fused);
    } else {
      hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, args, fused);
    }
    return hipGetLastError();
  };
  constexpr int R = kSoloRegisterSteps;
  if (solo_in_registers(all_blocks, args.n, layout, opt)) {
    if (layout == 1) return split ? go(&rollout_solo_kernel<1, true, R>, 3, 2 * kWave) : go(&rollout_solo_kernel<1, false, R>, 4, kWave);
    if (layout == 0) return split ? go(&rollout_solo_kernel<0, true, R>, 5, 2 * kWave) : go(&rollout_solo_kernel<0, false, R>, 6, kWave);
  }
  if (layout == 1) return split ? go(&rollout_solo_kernel<1, true, 0>, 7, 2 * kWave) : go(&rollout_solo_kernel<1, false, 0>, 8, kWave);
  if (layout == 0) return split ? go(&rollout_solo_kernel<0, true, 0>, 9, 2 * kWave) : go(&rollout_solo_kernel<0, false, 0>, 10, kWave);
  return hipErrorInvalidValue;
}

int softmin_chunks(int N) { return (N + kSoftChunk - 1) / kSoftChunk; }

hipError_t launch_softmin(int layout, const SoftminArgs& args, hipStream_t s) {
  clear_stale_error();
  const dim3 grid(args.chunks, args.P);
  if (layout == 0) {
    hipLaunchKernelGGL((softmin_partial_kernel<0>), grid, dim3(kSoftBlock), 0, s, args);
  } else if (layout == 1) {
    hipLaunchKernelGGL((softmin_partial_kernel<1>), grid, dim3(kSoftBlock), 0, s, args);
  } else {
    return hipErrorInvalidValue;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(softmin_final_kernel, dim3(args.P), dim3(kSoftBlock), 0, s, args);
  return hipGetLastError();
}
#endif  // ACMPC_TEMPORAL_TU

}  // namespace acmpc
