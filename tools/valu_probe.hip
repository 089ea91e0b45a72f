// Development probe: ISSUE cost per SIMD of the instructions the mode-T rollout is made of, at the occupancy the kernel
// itself runs at.  W workgroups of 256 threads per CU (W = 1, 2, 4, 7, 8 waves per SIMD; a dynamic LDS allocation of
// 160 KiB / W per workgroup keeps the dispatcher from placing more), every wave runs REPS x 32 independent instructions
// of one kind between two s_memtime stamps.  Reported per kind and W:
//     cycles per instruction per SIMD = median wave's stamp difference / (REPS x 32) / W
// (the W waves of a SIMD share its issue port), and the wall-clock ns per instruction and SIMD from the launch's event
// pair (which also sees the clock the chip actually ran at).  `--json` prints one JSON object (profiles/r04_valu_probe.json
// is that output); bench.py prices a kernel's opcode histogram with the W = 8 column (`roofline_valu`).
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe [--json]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

constexpr int kReps = 1000;

// VOP3 forms with three VGPR sources, VOP2 forms, literal forms, compares into SGPR pairs, LDS reads (per-lane gather of
// 16 bytes at 32-byte strides: the key table's pattern), scalar mask arithmetic
enum Kind {
  ADD, MUL, FMAC, SUB, XOR, LSHL, ADD_U32, MOV, MAX, MIN_I32,       // VOP2, two operands
  FMAMK, FMAAK,                                                      // VOP2 with a 32-bit literal
  FMA, MIN3, MED3, ADD3_U32, LSHL_ADD_U32, MAD_U64,                  // VOP3, three operands
  CMP_EQ_SGPR, CNDMASK_SGPR, SUB_ABS,                                // VOP3 encodings of two-operand work
  PK_MUL, PK_FMA,                                                    // packed
  DS_B128, DS_B96, DS_READ2_B32, DS_B64,                             // LDS gathers
  S_OR_B64, S_ANDN2_B64,                                             // scalar mask arithmetic
  LOG, SIN, SQRT, RCP,                                               // transcendental unit
  MIX_T,                                                             // the mode T step's own mix (see below)
  KINDS
};
static const char* kNames[KINDS] = {
    "v_add_f32", "v_mul_f32", "v_fmac_f32", "v_sub_f32", "v_xor_b32", "v_lshlrev_b32", "v_add_u32", "v_mov_b32", "v_max_f32",
    "v_min_i32", "v_fmamk_f32", "v_fmaak_f32", "v_fma_f32", "v_min3_f32", "v_med3_f32", "v_add3_u32", "v_lshl_add_u32",
    "v_mad_u64_u32", "v_cmp_eq_f32_e64", "v_cndmask_b32_e64", "v_sub_f32_e64_abs", "v_pk_mul_f32", "v_pk_fma_f32",
    "ds_read_b128", "ds_read_b96", "ds_read2_b32", "ds_read_b64", "s_or_b64", "s_andn2_b64", "v_log_f32", "v_sin_f32",
    "v_sqrt_f32", "v_rcp_f32", "mix_mode_T_step"};

#define VOP2_BODY(op)                                                                                         \
  asm volatile(REP8(op " %0, %4, %5\n " op " %1, %4, %6\n " op " %2, %5, %6\n " op " %3, %6, %6\n")         \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c))
#define VOP3_BODY(op)                                                                                                        \
  asm volatile(REP8(op " %0, %4, %5, %6\n " op " %1, %4, %6, %5\n " op " %2, %5, %6, %4\n " op " %3, %6, %6, %4\n")        \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c))
#define VOP1_BODY(op)                                                                                   \
  asm volatile(REP8(op " %0, %4\n " op " %1, %5\n " op " %2, %6\n " op " %3, %4\n")                   \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c))

template <int KIND>
__global__ void __launch_bounds__(256) probe(long long* cycles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int e = threadIdx.x; e < 2048; e += 256) lds[e] = static_cast<float>(e);
  __syncthreads();
  float a = threadIdx.x + 1.0f, b = 1.0f + 0.001f * threadIdx.x, c = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned long long u64;
  f2 pa = {a, b}, pb = {b, a}, pc = {c, c};
  float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  f2 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0};
  u64 m0 = 0, m1 = 0;
  f4 w4 = {0, 0, 0, 0};
  typedef float f3 __attribute__((ext_vector_type(3)));
  f3 w3 = {0, 0, 0};
  // per-lane gather: lanes spread over a few neighbouring 32-byte entries, as a wave's search windows are
  const unsigned gather = ((threadIdx.x * 7u) & 3u) * 32u;   // four neighbouring entries: conflict-free, like the kernel's
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < kReps; ++i) {
    if constexpr (KIND == ADD) VOP2_BODY("v_add_f32");
    else if constexpr (KIND == MUL) VOP2_BODY("v_mul_f32");
    else if constexpr (KIND == FMAC) VOP2_BODY("v_fmac_f32");
    else if constexpr (KIND == SUB) VOP2_BODY("v_sub_f32");
    else if constexpr (KIND == XOR) VOP2_BODY("v_xor_b32");
    else if constexpr (KIND == LSHL) VOP2_BODY("v_lshlrev_b32");
    else if constexpr (KIND == ADD_U32) VOP2_BODY("v_add_u32");
    else if constexpr (KIND == MAX) VOP2_BODY("v_max_f32");
    else if constexpr (KIND == MIN_I32) VOP2_BODY("v_min_i32");
    else if constexpr (KIND == MOV) VOP1_BODY("v_mov_b32");
    else if constexpr (KIND == LOG) VOP1_BODY("v_log_f32");
    else if constexpr (KIND == SIN) VOP1_BODY("v_sin_f32");
    else if constexpr (KIND == SQRT) VOP1_BODY("v_sqrt_f32");
    else if constexpr (KIND == RCP) VOP1_BODY("v_rcp_f32");
    else if constexpr (KIND == FMAMK) {
      asm volatile(REP8("v_fmamk_f32 %0, %4, 0x3ea2f983, %5\n v_fmamk_f32 %1, %4, 0x3ea2f983, %6\n v_fmamk_f32 %2, %5, 0x3ea2f983, %6\n v_fmamk_f32 %3, %6, 0x3ea2f983, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == FMAAK) {
      asm volatile(REP8("v_fmaak_f32 %0, %4, %5, 0x3c088734\n v_fmaak_f32 %1, %4, %6, 0x3c088734\n v_fmaak_f32 %2, %5, %6, 0x3c088734\n v_fmaak_f32 %3, %6, %4, 0x3c088734\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == FMA) VOP3_BODY("v_fma_f32");
    else if constexpr (KIND == MIN3) VOP3_BODY("v_min3_f32");
    else if constexpr (KIND == MED3) VOP3_BODY("v_med3_f32");
    else if constexpr (KIND == ADD3_U32) VOP3_BODY("v_add3_u32");
    else if constexpr (KIND == LSHL_ADD_U32) {
      asm volatile(REP8("v_lshl_add_u32 %0, %4, 5, %5\n v_lshl_add_u32 %1, %4, 5, %6\n v_lshl_add_u32 %2, %5, 5, %6\n v_lshl_add_u32 %3, %6, 5, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == MAD_U64) {
      asm volatile(REP8("v_mad_u64_u32 %0, s[10:11], %4, 12, %1\n v_mad_u64_u32 %2, s[12:13], %5, 12, %3\n v_mad_u64_u32 %0, s[10:11], %6, 12, %1\n v_mad_u64_u32 %2, s[12:13], %4, 12, %3\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a), "v"(b), "v"(c) : "s10", "s11", "s12", "s13");
    } else if constexpr (KIND == CMP_EQ_SGPR) {
      asm volatile(REP8("v_cmp_eq_f32_e64 s[10:11], %0, %1\n v_cmp_eq_f32_e64 s[12:13], %1, %0\n v_cmp_eq_f32_e64 s[14:15], %0, %0\n v_cmp_eq_f32_e64 s[16:17], %1, %1\n")
                   : : "v"(a), "v"(b) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
    } else if constexpr (KIND == CNDMASK_SGPR) {
      asm volatile(REP8("v_cndmask_b32_e64 %0, %4, %5, s[10:11]\n v_cndmask_b32_e64 %1, %4, %6, s[10:11]\n v_cndmask_b32_e64 %2, %5, %6, s[10:11]\n v_cndmask_b32_e64 %3, %6, %4, s[10:11]\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c) : "s10", "s11");
    } else if constexpr (KIND == SUB_ABS) {
      asm volatile(REP8("v_sub_f32_e64 %0, |%4|, %5\n v_sub_f32_e64 %1, |%4|, %6\n v_sub_f32_e64 %2, |%5|, %6\n v_sub_f32_e64 %3, |%6|, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == PK_MUL) {
      asm volatile(REP8("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %6\n v_pk_mul_f32 %2, %5, %6\n v_pk_mul_f32 %3, %6, %6\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(pa), "v"(pb), "v"(pc));
    } else if constexpr (KIND == PK_FMA) {
      asm volatile(REP8("v_pk_fma_f32 %0, %4, %5, %6\n v_pk_fma_f32 %1, %4, %6, %5\n v_pk_fma_f32 %2, %5, %6, %4\n v_pk_fma_f32 %3, %6, %6, %4\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(pa), "v"(pb), "v"(pc));
    } else if constexpr (KIND == DS_B128) {
      asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:64\n ds_read_b128 %0, %1 offset:128\n ds_read_b128 %0, %1 offset:16\n")
                   "s_waitcnt lgkmcnt(0)\n" : "+v"(w4) : "v"(gather));
    } else if constexpr (KIND == DS_B96) {
      asm volatile(REP8("ds_read_b96 %0, %1\n ds_read_b96 %0, %1 offset:64\n ds_read_b96 %0, %1 offset:128\n ds_read_b96 %0, %1 offset:16\n")
                   "s_waitcnt lgkmcnt(0)\n" : "+v"(w3) : "v"(gather));
    } else if constexpr (KIND == DS_READ2_B32) {
      asm volatile(REP8("ds_read2_b32 %0, %1 offset1:1\n ds_read2_b32 %0, %1 offset0:3 offset1:4\n ds_read2_b32 %0, %1 offset0:6 offset1:7\n ds_read2_b32 %0, %1 offset0:9 offset1:10\n")
                   "s_waitcnt lgkmcnt(0)\n" : "+v"(q0) : "v"(gather));
    } else if constexpr (KIND == DS_B64) {
      asm volatile(REP8("ds_read_b64 %0, %1\n ds_read_b64 %0, %1 offset:64\n ds_read_b64 %0, %1 offset:128\n ds_read_b64 %0, %1 offset:16\n")
                   "s_waitcnt lgkmcnt(0)\n" : "+v"(q0) : "v"(gather));
    } else if constexpr (KIND == S_OR_B64) {
      asm volatile(REP8("s_or_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n s_or_b64 %0, %0, %1\n s_or_b64 %1, %1, %0\n") : "+s"(m0), "+s"(m1) : : "scc");
    } else if constexpr (KIND == S_ANDN2_B64) {
      asm volatile(REP8("s_andn2_b64 %0, %0, %1\n s_andn2_b64 %1, %1, %0\n s_andn2_b64 %0, %0, %1\n s_andn2_b64 %1, %1, %0\n") : "+s"(m0), "+s"(m1) : : "scc");
    } else if constexpr (KIND == MIX_T) {
      // 32 vector instructions in the proportions of one mode T candidate-step with the 8-waypoint window (84 = 23 VOP2
      // two-operand, 9 literal FMAs, 8 v_fma + 4 min3/med3 + 3 integer VOP3, 8 compares into SGPRs, 3 selects ...), with 6 LDS
      // reads and 12 scalar mask operations between them: do the classes' costs simply add?
      asm volatile(
          REP4("v_fmac_f32 %0, %4, %5\n v_fma_f32 %1, %4, %6, %5\n v_fmac_f32 %2, %5, %6\n v_cmp_eq_f32_e64 s[10:11], %4, %5\n"
               "s_or_b64 s[12:13], s[10:11], s[12:13]\n v_fmaak_f32 %3, %6, %4, 0x3c088734\n v_mul_f32 %0, %4, %5\n"
               "v_sub_f32 %1, %4, %6\n s_andn2_b64 s[14:15], s[10:11], s[12:13]\n v_fma_f32 %2, %5, %6, %4\n")
          "v_min3_f32 %3, %6, %6, %4\n v_cndmask_b32_e64 %0, %4, %5, s[14:15]\n v_med3_f32 %1, %4, %6, %5\n v_add3_u32 %2, %5, %6, %4\n"
          : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c) : "s10", "s11", "s12", "s13", "s14", "s15", "scc");
      asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:64\n s_waitcnt lgkmcnt(0)\n" : "+v"(w4) : "v"(gather));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
  const float s = r0 + r1 + r2 + r3 + q0[0] + q0[1] + q1[0] + q1[1] + q2[0] + q2[1] + q3[0] + q3[1] + w4[0] + w4[3] + w3[0] + w3[2] +
                  static_cast<float>(m0 + m1);
  if (s == 123.456f) sink[0] = s + lds[0];
}

struct Result {
  double cycles_per_simd, ns_per_simd, kernel_ms;
};

template <int KIND>
Result run(int waves_per_simd, long long* d_cycles, float* d_sink, std::vector<long long>& h) {
  const int blocks = 256 * waves_per_simd;
  const size_t lds = ((160 * 1024 / waves_per_simd) / 1024) * 1024;   // exactly `waves_per_simd` workgroups fit a CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), lds, 0, d_cycles, d_sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), lds, 0, d_cycles, d_sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int waves = blocks * 4;
  hipMemcpy(h.data(), d_cycles, waves * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.begin() + waves);
  const double med = static_cast<double>(h[waves / 2]);
  // instructions of the measured class per loop trip: 32; the mix: 36 vector (its 8 scalar and 2 LDS riders come on top)
  const double per_trip = (KIND == MIX_T) ? 36.0 : 32.0;
  Result r;
  r.cycles_per_simd = med / (kReps * per_trip) / waves_per_simd;
  r.ns_per_simd = (ms * 1e6) / (kReps * per_trip * waves_per_simd);
  r.kernel_ms = ms;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return r;
}

template <int KIND>
void sweep(bool json, long long* d_cycles, float* d_sink, std::vector<long long>& h, std::string& out) {
  const int ws[5] = {1, 2, 4, 7, 8};
  char line[512];
  if (json) {
    std::snprintf(line, sizeof line, "%s  \"%s\": {", out.empty() ? "" : ",\n", kNames[KIND]);
    out += line;
  }
  for (int q = 0; q < 5; ++q) {
    const Result r = run<KIND>(ws[q], d_cycles, d_sink, h);
    if (json) {
      std::snprintf(line, sizeof line, "%s\"W%d\": {\"cycles_per_simd\": %.3f, \"ns_per_simd\": %.4f}", q ? ", " : "", ws[q],
                    r.cycles_per_simd, r.ns_per_simd);
      out += line;
    } else {
      std::printf("%-20s W=%d  cycles/instr/SIMD %7.3f   wall ns/instr/SIMD %6.3f  (kernel %.3f ms)\n", kNames[KIND], ws[q],
                  r.cycles_per_simd, r.ns_per_simd, r.kernel_ms);
    }
  }
  if (json) out += "}";
  if constexpr (KIND + 1 < KINDS) sweep<KIND + 1>(json, d_cycles, d_sink, h, out);
}

int main(int argc, char** argv) {
  const bool json = argc > 1 && std::strcmp(argv[1], "--json") == 0;
  long long* d_cycles;
  float* d_sink;
  hipMalloc(&d_cycles, 256 * 8 * 4 * sizeof(long long));
  hipMalloc(&d_sink, 64);
  std::vector<long long> h(256 * 8 * 4);
  std::string out;
  sweep<0>(json, d_cycles, d_sink, h, out);
  if (json) {
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    std::printf("{\n \"tool\": \"tools/valu_probe.hip --json\",\n \"unit\": \"s_memtime ticks per wave64 instruction per SIMD "
                "(median wave, W waves per SIMD share the port) and wall ns per instruction per SIMD\",\n"
                " \"device_clock_khz\": %d,\n \"instructions\": {\n%s\n }\n}\n", khz, out.c_str());
  }
  return 0;
}
