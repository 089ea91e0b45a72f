// Development probe: per-SIMD issue cost of the vector instructions the mode-T rollout is made of, at 1 / 2 / 4 waves
// per SIMD (one workgroup of 256 * W threads per CU, 256 workgroups).  Each wave runs REPS x 32 independent
// instructions of one kind and stamps s_memtime around the loop; cycles per instruction per SIMD =
// wave-cycles / (instructions * 1) / W-normalised (the W waves of a SIMD share its issue port).
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

constexpr int kReps = 2000;

enum Kind { ADD, FMA, PK_ADD, PK_MUL, PK_FMA, CNDMASK, CMP, MIN3, MAX3, MOV, FLOOR, ADD_LDS, PKFMA_LDS,
            CNDMASK_SGPR, CMP_CNDMASK, BFI, AND_OR, MIN3_U32, FMAC, MUL, XOR, CMP_SGPR, READ_B128, READ2_B32, BFE, MAXF, KINDS };
static const char* kNames[KINDS] = {"v_add_f32",     "v_fma_f32",  "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
                                    "v_cndmask_b32", "v_cmp_lt_f32", "v_min3_f32",  "v_max3_f32",   "v_mov_b32",
                                    "v_floor_f32",   "v_add+ds_read_b32 (1:1)", "v_pk_fma+ds_read_b64 (1:1)",
                                    "v_cndmask_b32_e64 sgpr", "v_cmp+v_cndmask pairs", "v_bfi_b32", "v_and_or_b32", "v_min3_u32",
                                    "v_fmac_f32", "v_mul_f32", "v_xor_b32", "v_cmp_lt_f32_e64 sgpr", "ds_read_b128 only", "ds_read2_b32 only", "v_bfe_i32", "v_max_f32"};

template <int KIND>
__global__ void probe(long long* cycles, float* sink) {
  __shared__ float lds[8192];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = threadIdx.x, b = 1.0f + threadIdx.x, c = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa = {a, b}, pb = {b, a}, pc = {c, c};
  float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  f2 q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0};
  const unsigned addr = (threadIdx.x & 255) * 4;
  const unsigned addr8 = (threadIdx.x & 255) * 8;
  const unsigned addr16 = (threadIdx.x & 255) * 16;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 w4 = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < kReps; ++i) {
    if constexpr (KIND == ADD) {
      asm volatile(REP8("v_add_f32 %0, %4, %5\n v_add_f32 %1, %4, %6\n v_add_f32 %2, %5, %6\n v_add_f32 %3, %6, %6\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == FMA) {
      asm volatile(REP8("v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %4, %6, %5\n v_fma_f32 %2, %5, %6, %4\n v_fma_f32 %3, %6, %6, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == PK_ADD) {
      asm volatile(REP8("v_pk_add_f32 %0, %4, %5\n v_pk_add_f32 %1, %4, %6\n v_pk_add_f32 %2, %5, %6\n v_pk_add_f32 %3, %6, %6\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(pa), "v"(pb), "v"(pc));
    } else if constexpr (KIND == PK_MUL) {
      asm volatile(REP8("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %6\n v_pk_mul_f32 %2, %5, %6\n v_pk_mul_f32 %3, %6, %6\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(pa), "v"(pb), "v"(pc));
    } else if constexpr (KIND == PK_FMA) {
      asm volatile(REP8("v_pk_fma_f32 %0, %4, %5, %6\n v_pk_fma_f32 %1, %4, %6, %5\n v_pk_fma_f32 %2, %5, %6, %4\n v_pk_fma_f32 %3, %6, %6, %4\n")
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(pa), "v"(pb), "v"(pc));
    } else if constexpr (KIND == CNDMASK) {
      asm volatile(REP8("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %6, vcc\n v_cndmask_b32 %2, %5, %6, vcc\n v_cndmask_b32 %3, %6, %4, vcc\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c) : "vcc");
    } else if constexpr (KIND == CMP) {
      asm volatile(REP32("v_cmp_lt_f32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
    } else if constexpr (KIND == MIN3) {
      asm volatile(REP8("v_min3_f32 %0, %4, %5, %6\n v_min3_f32 %1, %4, %6, %5\n v_min3_f32 %2, %5, %6, %4\n v_min3_f32 %3, %6, %6, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == MAX3) {
      asm volatile(REP8("v_max3_f32 %0, %4, %5, %6\n v_max3_f32 %1, %4, %6, %5\n v_max3_f32 %2, %5, %6, %4\n v_max3_f32 %3, %6, %6, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == MOV) {
      asm volatile(REP8("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == FLOOR) {
      asm volatile(REP8("v_floor_f32 %0, %4\n v_floor_f32 %1, %5\n v_floor_f32 %2, %6\n v_floor_f32 %3, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == ADD_LDS) {
      // 16 VALU + 16 LDS reads per group, waits only at the end of the group
      asm volatile(REP8("v_add_f32 %0, %4, %5\n ds_read_b32 %2, %7\n v_add_f32 %1, %4, %6\n ds_read_b32 %3, %7 offset:256\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(addr));
    } else if constexpr (KIND == PKFMA_LDS) {
      asm volatile(REP8("v_pk_fma_f32 %0, %4, %5, %6\n ds_read_b64 %2, %7\n v_pk_fma_f32 %1, %4, %6, %5\n ds_read_b64 %3, %7 offset:512\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(pa), "v"(pb), "v"(pc), "v"(addr8));
    } else if constexpr (KIND == CNDMASK_SGPR) {
      asm volatile(REP8("v_cndmask_b32_e64 %0, %4, %5, s[10:11]\n v_cndmask_b32_e64 %1, %4, %6, s[10:11]\n v_cndmask_b32_e64 %2, %5, %6, s[10:11]\n v_cndmask_b32_e64 %3, %6, %4, s[10:11]\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c) : "s10", "s11");
    } else if constexpr (KIND == CMP_CNDMASK) {
      // 16 compares + 16 selects, each select reading the mask the compare before it wrote (what compiled code does)
      asm volatile(REP8("v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %0, %4, %5, vcc\n v_cmp_lt_f32 vcc, %5, %6\n v_cndmask_b32 %1, %4, %6, vcc\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c) : "vcc");
    } else if constexpr (KIND == BFI) {
      asm volatile(REP8("v_bfi_b32 %0, %4, %5, %6\n v_bfi_b32 %1, %4, %6, %5\n v_bfi_b32 %2, %5, %6, %4\n v_bfi_b32 %3, %6, %6, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == AND_OR) {
      asm volatile(REP8("v_and_or_b32 %0, %4, %5, %6\n v_and_or_b32 %1, %4, %6, %5\n v_and_or_b32 %2, %5, %6, %4\n v_and_or_b32 %3, %6, %6, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == MIN3_U32) {
      asm volatile(REP8("v_min3_u32 %0, %4, %5, %6\n v_min3_u32 %1, %4, %6, %5\n v_min3_u32 %2, %5, %6, %4\n v_min3_u32 %3, %6, %6, %4\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == FMAC) {
      asm volatile(REP8("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %6\n v_fmac_f32 %2, %5, %6\n v_fmac_f32 %3, %6, %6\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == MUL) {
      asm volatile(REP8("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %6\n v_mul_f32 %2, %5, %6\n v_mul_f32 %3, %6, %6\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == XOR) {
      asm volatile(REP8("v_xor_b32 %0, %4, %5\n v_xor_b32 %1, %4, %6\n v_xor_b32 %2, %5, %6\n v_xor_b32 %3, %6, %6\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == CMP_SGPR) {
      asm volatile(REP8("v_cmp_lt_f32_e64 s[10:11], %0, %1\n v_cmp_lt_f32_e64 s[12:13], %1, %0\n v_cmp_lt_f32_e64 s[14:15], %0, %0\n v_cmp_lt_f32_e64 s[16:17], %1, %1\n")
                   : : "v"(a), "v"(b) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
    } else if constexpr (KIND == READ_B128) {
      asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:4096\n ds_read_b128 %0, %1 offset:8192\n ds_read_b128 %0, %1 offset:1024\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(w4) : "v"(addr16));
    } else if constexpr (KIND == READ2_B32) {
      asm volatile(REP8("ds_read2_b32 %0, %1 offset1:49\n ds_read2_b32 %0, %1 offset0:3 offset1:52\n ds_read2_b32 %0, %1 offset0:5 offset1:54\n ds_read2_b32 %0, %1 offset0:7 offset1:56\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(q0) : "v"(addr));
    } else if constexpr (KIND == BFE) {
      asm volatile(REP8("v_bfe_i32 %0, %4, 0, 1\n v_bfe_i32 %1, %5, 0, 1\n v_bfe_i32 %2, %6, 1, 1\n v_bfe_i32 %3, %4, 1, 1\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    } else if constexpr (KIND == MAXF) {
      asm volatile(REP8("v_max_f32 %0, %4, %5\n v_max_f32 %1, %4, %6\n v_max_f32 %2, %5, %6\n v_max_f32 %3, %6, %6\n")
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  const float s = r0 + r1 + r2 + r3 + q0[0] + q0[1] + q1[0] + q1[1] + q2[0] + q2[1] + q3[0] + q3[1] + w4[0] + w4[3];
  if (s == 123.456f) sink[0] = s + lds[0];
}

template <int KIND>
void run(int waves_per_simd, long long* d_cycles, float* d_sink, std::vector<long long>& h) {
  const int threads = 256 * waves_per_simd;
  const int blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int waves = blocks * threads / 64;
  hipMemcpy(h.data(), d_cycles, waves * sizeof(long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.begin() + waves);
  const double med = static_cast<double>(h[waves / 2]);
  // VALU instructions per loop iteration: 32, except the two mixed kinds (16 VALU + 16 LDS)
  const double valu = (KIND == ADD_LDS || KIND == PKFMA_LDS) ? 16.0 : 32.0  /* LDS-only kinds: 32 LDS instructions */;
  const double per_wave = med / (kReps * valu);
  // s_memtime ticks at 100 MHz on gfx950? -> report both raw ticks and wall-derived ns
  const double ns_per_instr_simd = (ms * 1e6) / (kReps * valu * waves_per_simd);
  std::printf("%-28s W=%d  memtime-ticks/instr/wave %7.3f  -> per SIMD %7.3f   wall ns/instr/SIMD %6.3f  (kernel %.3f ms)\n",
              kNames[KIND], waves_per_simd, per_wave, per_wave / waves_per_simd, ns_per_instr_simd, ms);
}

int main() {
  long long* d_cycles;
  float* d_sink;
  hipMalloc(&d_cycles, 256 * 32 * sizeof(long long));
  hipMalloc(&d_sink, 64);
  std::vector<long long> h(256 * 32);
  for (int w : {1, 2, 4, 7}) {
    run<ADD>(w, d_cycles, d_sink, h);
    run<FMA>(w, d_cycles, d_sink, h);
    run<PK_ADD>(w, d_cycles, d_sink, h);
    run<PK_MUL>(w, d_cycles, d_sink, h);
    run<PK_FMA>(w, d_cycles, d_sink, h);
    run<CNDMASK>(w, d_cycles, d_sink, h);
    run<CMP>(w, d_cycles, d_sink, h);
    run<MIN3>(w, d_cycles, d_sink, h);
    run<MAX3>(w, d_cycles, d_sink, h);
    run<MOV>(w, d_cycles, d_sink, h);
    run<FLOOR>(w, d_cycles, d_sink, h);
    run<ADD_LDS>(w, d_cycles, d_sink, h);
    run<PKFMA_LDS>(w, d_cycles, d_sink, h);
    run<CNDMASK_SGPR>(w, d_cycles, d_sink, h);
    run<CMP_CNDMASK>(w, d_cycles, d_sink, h);
    run<BFI>(w, d_cycles, d_sink, h);
    run<AND_OR>(w, d_cycles, d_sink, h);
    run<MIN3_U32>(w, d_cycles, d_sink, h);
    run<FMAC>(w, d_cycles, d_sink, h);
    run<MUL>(w, d_cycles, d_sink, h);
    run<XOR>(w, d_cycles, d_sink, h);
    run<CMP_SGPR>(w, d_cycles, d_sink, h);
    run<READ_B128>(w, d_cycles, d_sink, h);
    run<READ2_B32>(w, d_cycles, d_sink, h);
    run<BFE>(w, d_cycles, d_sink, h);
    run<MAXF>(w, d_cycles, d_sink, h);
  }
  return 0;
}
