// Micro-benchmark: VALU issue rate on gfx950, per instruction class and per occupancy (design input for the floor table).
//
// Question: does a wave64 VALU instruction occupy its SIMD for 4 cycles (what rounds 4-5 inferred from
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 quad-cycles) or for 2 (MI355X_MICROARCH.md: SIMD-32)?  Every wavefront runs ITER x 64
// instructions of ONE class on 8 independent register chains (no dependency closer than 8 instructions), timed with s_memtime
// (shader cycles) inside the kernel and with HIP events outside; w wavefronts per SIMD run the same stream side by side.
// Printed: cycles of the SIMD per wave-instruction = (t1 - t0) / (w x instructions per wave), and chip-wide G wave-instr/s.
//   build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { FMA, ADD_DPP, MOV_DPP, PK_FMA, PK_MUL, CVT_F32_F16, CVT_F16_F32, CVT_I32_F32, MUL_LO_U32, MAD_U64_U32, XOR, AND_OR, CNDMASK, CMP_CND,
          LSHL_B64, EXP, RCP, PK_MAX_F16, ADD_U32, ADD_CO_PAIR, BFE, FLOOR, FMA_DEP, ADD_DPP_DEP, MAX3, PERM, N_OPS };
static const char* op_names[N_OPS] = {
    "v_fma_f32", "v_add_f32 dpp row_shr:1", "v_mov_b32 dpp row_shr:1", "v_pk_fma_f32", "v_pk_mul_f32", "v_cvt_f32_f16", "v_cvt_f16_f32",
    "v_cvt_i32_f32", "v_mul_lo_u32", "v_mad_u64_u32", "v_xor_b32", "v_and_or_b32", "v_cndmask_b32 (vcc)", "v_cmp_gt_f32 + v_cndmask",
    "v_lshlrev_b64", "v_exp_f32", "v_rcp_f32", "v_pk_max_f16", "v_add_u32", "v_add_co_u32 + v_addc_co_u32", "v_bfe_u32", "v_floor_f32",
    "v_fma_f32 (dependent chain)", "v_add_f32 dpp (dependent chain)", "v_max3_f32", "v_perm_b32"};
// instructions issued per "slot" (CMP_CND and ADD_CO_PAIR issue two)
static const int op_instr[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1};

template <int OP>
__global__ void __launch_bounds__(1024) valu_kernel(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f + blockIdx.x * 1e-9f, c = 1e-7f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6}, pb = {b, b}, pc = {c, c};
  uint32_t u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, ub = 0x9E3779B1u + blockIdx.x;
  uint64_t q0 = u0, q1 = u1, q2 = u2, q3 = u3, q4 = u4, q5 = u5, q6 = u6, q7 = u7;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == ADD_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == MOV_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a##i) : "v"(b));
        REP8(X)
#undef X
      } else if (OP == PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(pb), "v"(pc));
        REP8(X)
#undef X
      } else if (OP == PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(pb));
        REP8(X)
#undef X
      } else if (OP == CVT_F32_F16) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (OP == CVT_F16_F32) {
#define X(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (OP == CVT_I32_F32) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (OP == MUL_LO_U32) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##i) : "v"(ub));
        REP8(X)
#undef X
      } else if (OP == MAD_U64_U32) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(u##i), "v"(ub) : "vcc");
        REP8(X)
#undef X
      } else if (OP == XOR) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u##i) : "v"(ub));
        REP8(X)
#undef X
      } else if (OP == AND_OR) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0));
        REP8(X)
#undef X
      } else if (OP == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub) : );
        REP8(X)
#undef X
      } else if (OP == CMP_CND) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
        REP8(X)
#undef X
      } else if (OP == LSHL_B64) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q##i));
        REP8(X)
#undef X
      } else if (OP == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (OP == RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (OP == PK_MAX_F16) {
#define X(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u##i) : "v"(ub));
        REP8(X)
#undef X
      } else if (OP == ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u##i) : "v"(ub));
        REP8(X)
#undef X
      } else if (OP == ADD_CO_PAIR) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(u##i), "+v"(a##i) : "v"(ub), "v"(u0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == BFE) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(u##i));
        REP8(X)
#undef X
      } else if (OP == FLOOR) {
#define X(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a##i));
        REP8(X)
#undef X
      } else if (OP == FMA_DEP) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == ADD_DPP_DEP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0));
        REP8(X)
#undef X
      } else if (OP == MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
        REP8(X)
#undef X
      } else if (OP == PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u##i) : "v"(ub), "v"(u0));
        REP8(X)
#undef X
      }
    }
  }
  const long long t1 = clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
  uint32_t us = u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);
  if (s == 123.456f || us == 0x12345u) out[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
static void run(float* out, long long* cyc, long long* hcyc) {
  const int iters = 2048;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  for (int w : {1, 2, 4, 8}) {  // wavefronts per SIMD
    const int threads = w <= 4 ? 256 * w : 1024;
    const int blocks = 256 * (w <= 4 ? 1 : w / 4);
    valu_kernel<OP><<<blocks, threads>>>(out, cyc, 16);
    hipDeviceSynchronize();
    hipEventRecord(s);
    valu_kernel<OP><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const int n_waves = blocks * threads / 64;
    hipMemcpy(hcyc, cyc, n_waves * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int i = 0; i < n_waves; ++i) { mean += (double)hcyc[i]; mx = hcyc[i] > mx ? (double)hcyc[i] : mx; }
    mean /= n_waves;
    const double n_instr = (double)iters * 64 * op_instr[OP];
    printf("%-34s w=%d waves/SIMD  %6.2f cycles/wave-instr (own stream)  %5.2f SIMD-cycles per wave-instr  %7.1f G wave-instr/s chip (%.3f ms, %.2f GHz eff.)\n",
           op_names[OP], w, mean / n_instr, mean / (n_instr * w), n_instr * n_waves / ms / 1e6, ms, mx / ms / 1e6);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096);
  hipMalloc(&cyc, 65536 * 8);
  long long* hcyc = (long long*)malloc(65536 * 8);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("# %s, %d CUs, clockRate %d kHz; clock64() = s_memtime\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
#define R(OPX) run<OPX>(out, cyc, hcyc);
  R(FMA) R(FMA_DEP) R(ADD_DPP) R(ADD_DPP_DEP) R(MOV_DPP) R(PK_FMA) R(PK_MUL) R(CVT_F32_F16) R(CVT_F16_F32) R(CVT_I32_F32) R(MUL_LO_U32) R(MAD_U64_U32)
  R(XOR) R(AND_OR) R(ADD_U32) R(ADD_CO_PAIR) R(BFE) R(PERM) R(CNDMASK) R(CMP_CND) R(LSHL_B64) R(EXP) R(RCP) R(FLOOR) R(PK_MAX_F16) R(MAX3)
  return 0;
}
