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

// Two x-neighbouring entries in one load.  On a hashed level with a power-of-two table the vertex (x, ...) and its neighbour
// (x + 1, ...) sit in entries i and i ^ 1 whenever x is even (the hash leaves the first coordinate unmultiplied: x ^ rest), i.e.
// in one aligned pair.  A random gather costs one slot of the address path per lane and line whatever it fetches
// (profiles/r02_ubench_gather.txt), so lanes whose cell has an even x issue 2^(D-1) loads instead of 2^D: a quarter of a
// level's gathers (and of its L2 misses) would go away -- MEASURED SLOWER (forward encode 6.24 -> 6.55 ms): the lanes with an
// odd x still need their two loads, so a wavefront issues three load instructions per corner pair instead of two, and the
// address path charges per instruction, not per active lane.  Off; kept for the record (tools/build_abl.sh -DL4D_PAIR_LOADS=1).
#ifndef L4D_PAIR_LOADS
#define L4D_PAIR_LOADS 0
#endif
template <int F>
struct PairVec;
template <>
struct PairVec<2> { typedef uint2 type; };
template <>
struct PairVec<4> { typedef uint4 type; };
template <>
struct PairVec<8> { typedef uint4 type; };  // unused: a pair of 16-byte entries is two loads anyway

// fp32-accumulated interpolation of one level; result NOT yet rounded.  PAIRS: the 3-D grids (the 2-D x time stacks have
// their own pair layout -- two time slices per entry -- and their fallback path has no register to spare).
template <int D, int F, bool PAIRS = (D == 3)>
__device__ __forceinline__ void level_lookup(const half_t* level_table, float scale, uint32_t res, uint32_t size,
                                             bool hashed, const float x[D], float out[F]) {
  Cell<D> c = locate<D>(x, scale);
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] = 0.0f;
  typedef typename EntryVec<F>::type EV;
  if (L4D_PAIR_LOADS && PAIRS && F <= 4 && hashed && (size & (size - 1)) == 0) {  // uniform over the level
    typedef typename PairVec<F>::type PV;
    const bool even = (c.cell[0] & 1u) == 0;
#pragma unroll
    for (int k = 0; k < (1 << D); k += 2) {  // corners k, k + 1 differ in x only; same accumulation order as below
      uint32_t g0[D], g1[D];
      const float w0 = corner<D>(c, k, g0), w1 = corner<D>(c, k + 1, g1);
      const uint32_t i0 = grid_index<D>(g0, res, size, true);
      EV e[2];
      if (even) {
        const PV r = *reinterpret_cast<const PV*>(level_table + (size_t)(i0 & ~1u) * F);
        const EV lo = *reinterpret_cast<const EV*>(&r), hi = *(reinterpret_cast<const EV*>(&r) + 1);
        const bool odd = i0 & 1u;  // selects, not an indexed register array (that would live in scratch)
        e[0] = odd ? hi : lo;
        e[1] = odd ? lo : hi;
      } else {
        e[0] = *reinterpret_cast<const EV*>(level_table + (size_t)i0 * F);
        e[1] = *reinterpret_cast<const EV*>(level_table + (size_t)grid_index<D>(g1, res, size, true) * F);
      }
      const half_t* h0 = reinterpret_cast<const half_t*>(&e[0]);
      const half_t* h1 = reinterpret_cast<const half_t*>(&e[1]);
#pragma unroll
      for (int f = 0; f < F; ++f) out[f] = fmix(h0[f], w0, out[f]);
#pragma unroll
      for (int f = 0; f < F; ++f) out[f] = fmix(h1[f], w1, out[f]);
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

