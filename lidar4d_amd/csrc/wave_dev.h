// Wave64 cooperative helpers for gradient accumulation on gfx950.
//
// Measured on MI355X (profiles/r01_ubench_*.txt): scattered global fp32 atomics sustain ~20 G lane-ops/s no matter
// the scope, coalesced ones 267 G/s; LDS integer atomics (ds_add_u32/u64) 4.8 / 2.9 T lane-ops/s, but LDS *float*
// atomics (ds_add_f32, ds_pk_add_f16) only 0.2 T/s.  Hence: (1) pre-reduce runs of equal addresses inside a wave
// before touching global memory, (2) accumulate in LDS as fixed-point integers (which is also order-independent,
// i.e. deterministic), flush once, coalesced.
#pragma once
#include "common.h"

// Run-length pre-reduction: consecutive lanes (consecutive samples of a ray) that target the same key are summed
// into the first lane of the run.  Returns true for lanes that must issue the atomic (run heads; every active lane
// when the wave's keys are mostly distinct and the shuffle work would be wasted).  All 64 lanes must call this.
// max_heads: merge only when the wave holds at most this many runs.  For global atomics (20 G/s) merging nearly always
// pays (default 40); in front of LDS integer atomics a same-address pile-up only costs ~1 cycle per lane, so the
// 6*K shuffles are worth it for very long runs only (pass 4).
template <int K>
__device__ __forceinline__ bool wave_run_reduce(uint32_t key, bool active, float v[K], int max_heads = 40) {
  const int lane = __lane_id();
  if (!active) key = 0xFFFFFF00u | (uint32_t)lane;  // unique, never merged
  const uint32_t prev = __shfl_up(key, 1, 64);
  const bool head = (lane == 0) || (key != prev);
  const unsigned long long H = __ballot(head);
  if (__popcll(H) > max_heads) return active;  // wave-uniform: little to merge
  const unsigned long long above = (lane == 63) ? 0ull : (H & ~((2ull << lane) - 1ull));
  const int end = above ? (__ffsll((long long)above) - 2) : 63;  // last lane of this lane's run
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float t = __shfl_down(v[k], d, 64);
      if (lane + d <= end) v[k] += t;
    }
  }
  return head && active;
}

// Run merge inside DPP rows.  Consecutive lanes are consecutive samples of a ray, so lanes that target the same texel /
// table entry form contiguous runs.  row_runs() finds the runs inside each 16-lane row from the keys (one row_shr
// compare + one ballot), row_scan() is a segmented Hillis-Steele scan over them -- per value 4 v_fmac_f32_dpp (pure
// VALU; the flags f1..f8 switch the adds off across run boundaries) -- after which the LAST lane of every run holds the
// run's total and is the only one that needs to issue an atomic.  Rows that hold several runs (fine scales, where 16
// samples cross 2-3 texels) are merged run by run.
struct RowRuns {
  float f1, f2, f4, f8;  // 1.0 if the lane 1/2/4/8 to the left is in the same run (and the same row), else 0.0
  bool tail;             // this lane is the last of its run
};
// n_heads (optional): number of runs in the wave (64 = nothing to merge)
// WIDTH (16 / 8 / 4): runs are additionally cut at multiples of WIDTH lanes, so that row_scan needs log2(WIDTH) steps only
// (fewer v_fmac_dpp per value, more run tails, i.e. more LDS atomics: a trade the VALU-bound plane adjoints can choose)
template <int WIDTH = 16>
__device__ __forceinline__ RowRuns row_runs(uint32_t key, int* n_heads = nullptr) {
  const int lane = __lane_id();
  // old = ~key with bound_ctrl off: the first lane of a row keeps ~key, i.e. always starts a run
  const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x111 /*row_shr:1*/, 0xf, 0xf, false);
  unsigned long long H = __ballot(key != prev);                        // run heads
  if (WIDTH == 8) H |= 0x0101010101010101ull;
  if (WIDTH == 4) H |= 0x1111111111111111ull;
  if (n_heads) *n_heads = __popcll(H);
  const unsigned long long le = H & (~0ull >> (63 - lane));            // heads at or below this lane (never empty)
  const int off = lane - (63 - __clzll((long long)le));                // distance to this lane's run head, 0..15
  RowRuns r;
  r.f1 = off >= 1 ? 1.0f : 0.0f;
  r.f2 = off >= 2 ? 1.0f : 0.0f;
  r.f4 = off >= 4 ? 1.0f : 0.0f;
  r.f8 = off >= 8 ? 1.0f : 0.0f;
  r.tail = lane == 63 || ((H >> (lane + 1)) & 1ull) != 0ull;
  return r;
}
template <int K, int WIDTH = 16>
__device__ __forceinline__ void row_scan(const RowRuns& r, float v[K]) {
  // v += flag * dpp_row_shr(v) as ONE instruction.  (hipcc keeps a v_mov_b32_dpp + v_fmac pair for the builtin form.)
  // Inline asm is invisible to the hazard recogniser: an EXEC write needs 5 wait states before a DPP op and a VALU write
  // of a DPP source 2, so the sequence starts with s_nop 4, and the steps run value-major (8 independent instructions
  // between a write of v[k] and its next DPP read).
#define L4D_FMAC_DPP(x, f, shr) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:" #shr " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(f))
  asm volatile("s_nop 4");
#pragma unroll
  for (int k = 0; k < K; ++k) L4D_FMAC_DPP(v[k], r.f1, 1);
  if (K < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int k = 0; k < K; ++k) L4D_FMAC_DPP(v[k], r.f2, 2);
  if (WIDTH > 4) {
    if (K < 3) asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < K; ++k) L4D_FMAC_DPP(v[k], r.f4, 4);
  }
  if (WIDTH > 8) {
    if (K < 3) asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < K; ++k) L4D_FMAC_DPP(v[k], r.f8, 8);
  }
#undef L4D_FMAC_DPP
}

// Run merge over the WHOLE wavefront: the row scan above, then the row totals carried into runs that reach back over the start of
// their row -- three more v_fmac_f32_dpp per value (row_bcast:15 hands lane 15 of a row to every lane of the next one; row_mask
// picks the row that takes it, so rows 1, 2, 3 follow each other and a run that covers several rows collects all of them).  Seven
// DPP instructions per value and no LDS, where wave_run_reduce() needs six ds_bpermute round trips per value; the run's total ends
// up in its LAST lane.  heads: ballot of "this lane starts a run" over the wavefront (bit 0 is always set).
struct WaveRuns {
  RowRuns row;  // the row-local part (runs cut at multiples of 16 lanes)
  float fc;     // 1.0 if the lane's run started in an earlier row
  bool tail;    // this lane is the last of its run
};
__device__ __forceinline__ WaveRuns wave_runs(unsigned long long heads, int* n_heads = nullptr) {
  const int lane = __lane_id();
  heads |= 1ull;
  if (n_heads) *n_heads = __popcll(heads);
  const unsigned long long H = heads | 0x0001000100010001ull;          // row-local runs
  const unsigned long long le = H & (~0ull >> (63 - lane));
  const int off = lane - (63 - __clzll((long long)le));
  WaveRuns r;
  r.row.f1 = off >= 1 ? 1.0f : 0.0f;
  r.row.f2 = off >= 2 ? 1.0f : 0.0f;
  r.row.f4 = off >= 4 ? 1.0f : 0.0f;
  r.row.f8 = off >= 8 ? 1.0f : 0.0f;
  r.tail = lane == 63 || ((heads >> (lane + 1)) & 1ull) != 0ull;
  r.row.tail = r.tail;
  // no head between the start of this lane's row and the lane itself: the run comes from the row before
  const unsigned long long in_row = (heads >> (lane & 48)) & (0xFFFFull >> (15 - (lane & 15)));
  r.fc = in_row == 0ull ? 1.0f : 0.0f;
  return r;
}
template <int K>
__device__ __forceinline__ void wave_scan(const WaveRuns& r, float v[K]) {
  row_scan<K, 16>(r.row, v);
#define L4D_FMAC_BCAST(x, f, rmask) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:" #rmask " bank_mask:0xf" : "+v"(x) : "v"(f))
  if (K < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int k = 0; k < K; ++k) L4D_FMAC_BCAST(v[k], r.fc, 0x2);
  if (K < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int k = 0; k < K; ++k) L4D_FMAC_BCAST(v[k], r.fc, 0x4);
  if (K < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int k = 0; k < K; ++k) L4D_FMAC_BCAST(v[k], r.fc, 0x8);
#undef L4D_FMAC_BCAST
}

// running maximum of |v| that turns into +inf as soon as a non-finite value is seen (fmaxf alone drops nan): the
// fixed-point statistics double as the "gradient overflowed" signal of the adjoint chain (common.h, f2h_grad)
__device__ __forceinline__ float amax_nf(float m, float v) { return nonfinite(v) ? __builtin_inff() : fmaxf(m, fabsf(v)); }

// float -> fixed point for the integer LDS accumulators: floor(x + 0.5) as ONE instruction (v_cvt_rpi_i32_f32), where
// __float2int_rn is v_rndne_f32 + v_cvt_i32_f32.  The adjoint kernels convert every contribution they accumulate (576 per
// sample in the time-plane kernel alone) and are bound by VALU issue; exact ties round up instead of to even, which changes
// nothing that is measurable (the quantum is 2^-26 ... 2^-30 of the largest gradient) and keeps sums exact and order-independent.
__device__ __forceinline__ int fx_round(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// fixed-point scale: largest power of two s with bound * s < 2^bits (bound > 0)
__device__ __forceinline__ float fx_scale(float bound, int bits) {
  if (!(bound > 0.0f)) return 1.0f;
  int e;
  frexpf(bound, &e);  // bound < 2^e
  return ldexpf(1.0f, bits - e);
}

// Maximum over the wave of NON-NEGATIVE values, as a wave-uniform result.  DPP only (v_max_f32 with row_shr 1/2/4/8,
// then row_bcast:15 / row_bcast:31 carry the row maxima upwards; lane 63 ends up with the total) -- __shfl_xor would be
// six trips through the LDS crossbar.  old = 0 for lanes a DPP step does not reach: neutral for non-negative inputs.
__device__ __forceinline__ float wave_max(float v) {
#define L4D_MAX_DPP(ctrl, rmask) \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))
  L4D_MAX_DPP(0x111, 0xf);  // row_shr:1
  L4D_MAX_DPP(0x112, 0xf);  // row_shr:2
  L4D_MAX_DPP(0x114, 0xf);  // row_shr:4
  L4D_MAX_DPP(0x118, 0xf);  // row_shr:8   -> lane 15 of each row holds the row maximum
  L4D_MAX_DPP(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  L4D_MAX_DPP(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef L4D_MAX_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// non-negative floats order like their bit patterns
// The maxima are touched by every wave of a launch: read first (L2-scope load, cheap) and only issue the atomic when it
// can raise the value -- after the first few workgroups almost nobody does.
__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
  const unsigned int cur = __hip_atomic_load(reinterpret_cast<unsigned int*>(addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__float_as_uint(v) > cur) atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
