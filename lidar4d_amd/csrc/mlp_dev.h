// Device pieces of the register-resident MFMA chain (mlp.hip header comment: operand layouts, the permuted weight-row order that
// makes a layer's C fragment the next layer's B fragment), shared by mlp.hip and by the forward encode kernel (fused.hip), whose
// epilogue runs the density network's forward pass on the rows it has just staged in LDS.
#pragma once
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define HID 64
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int perm_row(int mt, int i) { return 32 * (mt >> 1) + 8 * (i >> 2) + 4 * (mt & 1) + (i & 3); }

// ---- fragment construction (once per block, into LDS) -----------------------------------------
// value(row, k) of an A/B fragment: lane (i, g), element e -> k = 32*ks + 8g + e
// kind 0: W[row_of(i)][k]          (forward A: rows = neurons of this layer, k = its inputs)
// kind 1: W[k][row_of(i)]          (W^T: rows = inputs of the layer, k = its neurons)
// cperm >= 0 (the attribute networks fed from AttrSrc): the kernel's PHYSICAL input column order differs from the weight
// matrix' logical one so that the geo features can be taken from the sigma network's output row [h0, g0 .. g14] with two
// aligned 16-byte loads: physical column cperm carries the constant 1.0 (logical column cperm + 15, the first padding
// column), physical cperm + 1 .. cperm + 15 carry g0 .. g14 (logical cperm .. cperm + 14); all other columns coincide.
__device__ __forceinline__ int col_map(int c, int cperm) {
  if (cperm < 0 || c < cperm || c >= cperm + 16) return c;
  return c == cperm ? cperm + 15 : c - 1;
}
__device__ __forceinline__ h8 build_frag(const half_t* __restrict__ W, int R, int Cw, int kind, int row, int kbase, int cperm = -1) {
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kbase + e;
    float x = 0.0f;
    if (kind == 0) {
      if (row < R && k < Cw) x = h2f(W[row * Cw + col_map(k, cperm)]);
    } else {
      if (k < R && row < Cw) x = h2f(W[k * Cw + col_map(row, cperm)]);
    }
    v[e] = f2h(x);
  }
  return v;
}

__device__ __forceinline__ h8 ident_frag(int lane, int half_sel) {
  const int j = lane & 15, g = lane >> 4;
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (8 * g + e == 16 * half_sel + j) ? (half_t)1.0f : (half_t)0.0f;
  return v;
}

__device__ __forceinline__ h8 relu_pack(const f4& lo, const f4& hi) {
  h8 v;
  // Convert first, then ONE packed fp16 maximum per pair (v_cvt_pk_f16_f32 + v_pk_max_f16: half the instructions of eight
  // v_max_f32 + four conversions; same values: rounding is monotone and keeps zero).  Measured over the step's MLP kernels:
  // -0.30 ms (gpurun_out/r4j).  The integer form (max of the bit patterns as int16) made the compiler split the conversions
  // again and lost most of that (r4k).  A result of -0 (from a tiny negative input) is possible here; every consumer compares
  // "> 0" as a float, for which it is a zero.
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 z = {(half_t)0.0f, (half_t)0.0f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    h2 a = {f2h(lo[2 * q]), f2h(lo[2 * q + 1])}, b = {f2h(hi[2 * q]), f2h(hi[2 * q + 1])};
    a = __builtin_elementwise_max(a, z);
    b = __builtin_elementwise_max(b, z);
    v[2 * q] = a[0]; v[2 * q + 1] = a[1];
    v[4 + 2 * q] = b[0]; v[4 + 2 * q + 1] = b[1];
  }
  return v;
}

// ReLU' on packed halfs: v's half survives where the same half of h -- a ReLU OUTPUT of relu_pack: never negative, and +0 where the
// input was -0 (v_pk_max_f16 orders -0 below +0: tools/ubench/pk_max_zero.hip) -- is non-zero.  min(h, 1) on the bit patterns is 0 / 1,
// and an integer multiply by it keeps or clears the half: TWO instructions per pair of elements, written as asm because the compiler turns
// any C form back into a compare + select per element (round 5: mixed results for that reason).  Per 32-row tile of the attribute
// backward: 128 v_cmp + 128 v_cndmask + 64 single conversions -> 64 v_pk_min_u16 + 64 v_pk_mul_lo_u16 + 32 more v_cvt_pk.
// RELU_GATE_ASM 0: the compare form (A/B).
#ifndef RELU_GATE_ASM
#define RELU_GATE_ASM 1
#endif
__device__ __forceinline__ h8 relu_gate(const f4& lo, const f4& hi, const h8& h) {  // halfs of (lo | hi), gated by h
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
  typedef float ff2 __attribute__((ext_vector_type(2)));
  // (plain dwords, each built from its own pair of floats -- one v_cvt_pk_f16_f32: with the eight halfs taken out of an h8 the compiler
  // converted them one by one and packed them with v_perm_b32; with the dwords as elements of a 4-vector it multiplied dwords 1 .. 3 by
  // the RESULT of dword 0, tools/ubench/relu_gate_check.hip)
  uint32_t hw[4], vw[4];
  __builtin_memcpy(hw, &h, 16);
  const uint32_t one = 0x00010001u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4& src = q < 2 ? lo : hi;
    const ff2 pair = {src[2 * (q & 1)], src[2 * (q & 1) + 1]};
    const hh2 half2 = __builtin_convertvector(pair, hh2);  // (= f2h_grad of both: round to nearest even, inf kept)
    // The minimum is asm (opaque: as C the pair min + multiply is folded back into compare + select); the MULTIPLY is C, so that the
    // instruction that defines an MFMA operand is one the compiler's hazard recogniser sees.
    uint32_t m;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(hw[q]), "s"(one));
    vw[q] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, half2) * __builtin_bit_cast(us2, m));
  }
  h8 out;
  __builtin_memcpy(&out, vw, 16);
  return out;
}
__device__ __forceinline__ float clamp_h(float x) { return fminf(fmaxf(x, -65504.0f), 65504.0f); }

