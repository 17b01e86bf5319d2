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

// (ReLU masks of the backward as bit-pattern operations on the packed halfs: measured mixed -- attribute backward 2.82 -> 2.71 ms, flow
// backward 1.26 -> 1.30 -- because the compiler turns them back into compares; removed in round 5.)
__device__ __forceinline__ float clamp_h(float x) { return fminf(fmaxf(x, -65504.0f), 65504.0f); }

