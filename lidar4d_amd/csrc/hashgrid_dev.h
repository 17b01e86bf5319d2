// Device functions of the multi-resolution hash-grid lookup (tiny-cuda-nn "HashGrid" semantics,
// SURVEY.md A.1), shared by hashgrid.hip (operator-level kernels) and fused.hip (per-point field kernel).
#pragma once
#include "common.h"

#define PRIME1 2654435761u
#define PRIME2 805459861u

template <int F>
struct EntryVec;
template <>
struct EntryVec<2> { typedef uint32_t type; };
template <>
struct EntryVec<4> { typedef uint2 type; };
template <>
struct EntryVec<8> { typedef uint4 type; };

template <int F>
__device__ __forceinline__ void load_entry(const half_t* p, float v[F]) {
  typename EntryVec<F>::type raw = *reinterpret_cast<const typename EntryVec<F>::type*>(p);
  const half_t* h = reinterpret_cast<const half_t*>(&raw);
#pragma unroll
  for (int f = 0; f < F; ++f) v[f] = h2f(h[f]);
}

// Entry index of grid vertex g (tiny-cuda-nn grid_index()).
template <int D>
__device__ __forceinline__ uint32_t grid_index(const uint32_t g[D], uint32_t res, uint32_t size, bool hashed) {
  uint32_t idx;
  if (hashed) {
    idx = g[0];
    if (D > 1) idx ^= g[1] * PRIME1;
    if (D > 2) idx ^= g[2] * PRIME2;
  } else {
    uint32_t stride = 1;
    idx = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (stride <= size) {
        idx += g[d] * stride;
        stride *= res;
      }
    }
  }
  // size is a power of two on every hashed level with 2^log2_hashmap_size entries
  return ((size & (size - 1)) == 0) ? (idx & (size - 1)) : (idx % size);
}

template <int D>
struct Cell {
  uint32_t cell[D];
  float frac[D];
};

template <int D>
__device__ __forceinline__ Cell<D> locate(const float x[D], float scale) {
  Cell<D> c;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float pos = fmaf(scale, x[d], 0.5f);
    float fl = floorf(pos);
    c.cell[d] = (uint32_t)(int)fl;  // negatives wrap like tiny-cuda-nn
    c.frac[d] = pos - fl;
  }
  return c;
}

template <int D>
__device__ __forceinline__ float corner(const Cell<D>& c, int corner_id, uint32_t g[D]) {
  float w = 1.0f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if ((corner_id >> d) & 1) {
      w *= c.frac[d];
      g[d] = c.cell[d] + 1u;
    } else {
      w *= 1.0f - c.frac[d];
      g[d] = c.cell[d];
    }
  }
  return w;
}

// fp32-accumulated interpolation of one level; result NOT yet rounded.
template <int D, int F>
__device__ __forceinline__ void level_lookup(const half_t* level_table, float scale, uint32_t res, uint32_t size,
                                             bool hashed, const float x[D], float out[F]) {
  Cell<D> c = locate<D>(x, scale);
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] = 0.0f;
#pragma unroll
  for (int k = 0; k < (1 << D); ++k) {
    uint32_t g[D];
    float w = corner<D>(c, k, g);
    uint32_t idx = grid_index<D>(g, res, size, hashed);
    float v[F];
    load_entry<F>(level_table + (size_t)idx * F, v);
    if (F % 2 == 0) {
#pragma unroll
      for (int f = 0; f + 1 < F; f += 2) {  // v_pk_mul_f32 + v_pk_add_f32: two features per instruction, same roundings
        const float2_t r = float2_t{out[f], out[f + 1]} + float2_t{v[f], v[f + 1]} * w;
        out[f] = r[0];
        out[f + 1] = r[1];
      }
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) out[f] += w * v[f];
    }
  }
}

