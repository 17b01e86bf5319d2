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

// acc + (float)h * w as ONE v_fma_mix_f32: the fp16 source is converted on the fly and the product is not rounded before the
// add (v_cvt + v_pk_mul + v_pk_add took 2 instructions per feature, this takes 1: the lookup kernels that run from LDS are
// bound by exactly these).  The result differs from the two-rounding form by < 1 fp32 ulp per corner, i.e. a rare fp16 ulp
// after the encoder's output rounding.
__device__ __forceinline__ float fmix(half_t h, float w, float acc) { return __builtin_fmaf((float)h, w, acc); }

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

// Hashed level with a power-of-two table (every hashed level tiny-cuda-nn's host code produces: n = 2^log2_hashmap_size): no
// stride walk, no modulo, no per-call decision -- callers test (hashed && size power of two) ONCE per level / launch and take
// this path in their inner loops (the generic form above compiles both addressings and a uniform branch into every lookup).
template <int D>
__device__ __forceinline__ uint32_t grid_index_fast(const uint32_t g[D], uint32_t mask) {
  uint32_t idx = g[0];
  if (D > 1) idx ^= g[1] * PRIME1;
  if (D > 2) idx ^= g[2] * PRIME2;
  return idx & mask;
}
__host__ __device__ __forceinline__ bool is_pow2(uint32_t v) { return v != 0u && (v & (v - 1u)) == 0u; }

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

// Two x-neighbouring entries per load where they share an aligned pair.  On a hashed level with a power-of-two table the vertex
// (x, ...) and its neighbour (x + 1, ...) sit in entries i and i ^ m, m = 2^(trailing ones of x + 1) - 1 (the hash leaves the first
// coordinate unmultiplied: x ^ rest): for an even x that is the aligned pair (i, i ^ 1) = ONE 16-byte load.  A gather costs the
// address path ~2.1 clocks per distinct address of an instruction whatever its width (tools/ubench/gather_policy.hip), so PAIRLD:
// every lane loads the 16-byte pair that holds its first corner (all lanes: one instruction), and only the lanes with an odd x
// issue a second, 8-byte load for the neighbour: 1.5 address slots per corner pair on average instead of 2.  (Round 2 tried the
// divergent form -- even lanes one 16-byte load, odd lanes two 8-byte loads: three instructions per corner pair -- inside the
// fused encode kernel and lost, 6.24 -> 6.55 ms.)  Same accumulation order as the plain path: bit-identical results.
// fp32-accumulated interpolation of one level; result NOT yet rounded.  PAIRS: the 3-D grids (the 2-D x time stacks have
// their own pair layout -- two time slices per entry -- and their fallback path has no register to spare).
template <int D, int F, bool PAIRLD = false>
__device__ __forceinline__ void level_lookup(const half_t* level_table, float scale, uint32_t res, uint32_t size,
                                             bool hashed, const float x[D], float out[F]) {
  Cell<D> c = locate<D>(x, scale);
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] = 0.0f;
  typedef typename EntryVec<F>::type EV;
  if (PAIRLD && F == 4 && hashed && (size & (size - 1)) == 0) {  // uniform over the level
    constexpr int NP = 1 << (D - 1);
    const bool odd_x = (c.cell[0] & 1u) != 0;
    float w0[NP], w1[NP];
    uint32_t i0[NP];
    uint4 pr[NP];
    uint2 nb[NP];
    // all requests first, unconditionally (a load under a condition is waited for where it is issued): the lanes with an even
    // x request entry 0 as their "neighbour" -- ONE address for all of them, i.e. one slot of the address path
#pragma unroll
    for (int q = 0; q < NP; ++q) {  // corners 2q, 2q + 1 differ in x only
      uint32_t g0[D], g1[D];
      w0[q] = corner<D>(c, 2 * q, g0);
      w1[q] = corner<D>(c, 2 * q + 1, g1);
      i0[q] = grid_index_fast<D>(g0, size - 1u);
      const uint32_t i1 = odd_x ? grid_index_fast<D>(g1, size - 1u) : 0u;
      pr[q] = *reinterpret_cast<const uint4*>(level_table + (size_t)(i0[q] & ~1u) * 4);  // entries (i0 & ~1), (i0 | 1)
      nb[q] = *reinterpret_cast<const uint2*>(level_table + (size_t)i1 * 4);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {  // same accumulation order as the plain path
      const bool hi0 = i0[q] & 1u;
      const uint2 e0 = hi0 ? make_uint2(pr[q].z, pr[q].w) : make_uint2(pr[q].x, pr[q].y);
      const uint2 pe = hi0 ? make_uint2(pr[q].x, pr[q].y) : make_uint2(pr[q].z, pr[q].w);  // even x: the neighbour is i0 ^ 1
      const uint2 e1 = odd_x ? nb[q] : pe;
      const half_t* h0 = reinterpret_cast<const half_t*>(&e0);
      const half_t* h1 = reinterpret_cast<const half_t*>(&e1);
#pragma unroll
      for (int f = 0; f < F; ++f) out[f] = fmix(h0[f], w0[q], out[f]);
#pragma unroll
      for (int f = 0; f < F; ++f) out[f] = fmix(h1[f], w1[q], out[f]);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < (1 << D); ++k) {
    uint32_t g[D];
    float w = corner<D>(c, k, g);
    uint32_t idx = grid_index<D>(g, res, size, hashed);
    const EV raw = *reinterpret_cast<const EV*>(level_table + (size_t)idx * F);
    const half_t* h = reinterpret_cast<const half_t*>(&raw);
#pragma unroll
    for (int f = 0; f < F; ++f) out[f] = fmix(h[f], w, out[f]);
  }
}

