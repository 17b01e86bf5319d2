// Shared descriptors/helpers of the fused per-point field kernels (fused.hip forward, field_bwd.hip backward).
#pragma once
#include "hashgrid_dev.h"
#include "planes_dev.h"

#define MAX_SLICES 8

struct FieldDesc {
  GridDesc hs;               // static 3-D hash grid, F = 4
  const half_t* hs_table;
  GridDesc hd[3];            // xy / xz / yz 2-D grids, F = 4
  const half_t* hd_tables[3][MAX_SLICES];
  // Pair-interleaved copies of the time-slice tables (or null): hd_pairs[plane][q][entry] = {slice q: 4 halfs, slice q+1:
  // 4 halfs}, q < n_slices - 1.  HashGridT blends two ADJACENT slices (hash_field.py:79-85), so one 16-byte load per
  // corner brings both -- half as many gather instructions as two 8-byte loads from two tables, with the same cache
  // footprint (the slice pair in use).  hd_entries[plane] = entries of one slice table (all levels).
  const half_t* hd_pairs[3];
  uint32_t hd_entries[3];
  int n_slices;
  PlaneDesc planes;
  const float* planes_cl;
};

struct FieldGrads {
  float* hs_table;
  float* hd_tables[3][MAX_SLICES];
  float* planes_cl;
};

struct TimeCoef {
  SlicePair sp;
  float basis[4];
};
__device__ __forceinline__ TimeCoef time_coef(float t, int n_slices) {
  TimeCoef c;
  c.sp = slice_pair(t, n_slices);
  lagrange4(t, c.basis);
  return c;
}

// pair table and half that hold the slice(s) of a time coefficient: pair q = min(i1, n_slices - 2); a single slice
// (i1 == i2) sits in the low half of pair i1, except the last slice: high half of the last pair
struct PairSel {
  int q;
  bool hi;
};
__device__ __forceinline__ PairSel pair_sel(const SlicePair& sp, int n_slices) {
  PairSel s;
  s.q = min(sp.i1, n_slices - 2);
  s.hi = sp.i1 > s.q;
  return s;
}

// ---- pair-table lookups in two steps: fetch the four corner entries of a cell, then interpolate / blend -------------
// The three frames of a sample (the point itself and its two flow-warped copies) are looked up in the same tables; where
// the warped point falls into the same cell as the point itself -- the flow is zero or tiny wherever the scene is static,
// which is most of it -- its four corner entries are the ones already fetched and only the weights differ.
struct PairCorners {
  uint4 e[4];  // corner k: {slice q: 4 halfs, slice q + 1: 4 halfs}
};
// (float)half * w + 0 as ONE v_fma_mix_f32 with the half taken from the low / high 16 bits of a packed word.  For the first
// corner of an interpolation (accumulator still zero) the compiler turns fmix(h, w, 0) into a conversion plus a multiply -- two
// instructions for a quarter of the multiply-adds of kernels that are bound by exactly those (dynhash_fwd_lds_kernel: VALU-bound).
// Same value as the product (a zero result may differ in sign, which no consumer sees).
template <bool HI>
__device__ __forceinline__ float fmix0(uint32_t word, float w) {
  float r;
  if (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(word), "v"(w));
  else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(word), "v"(w));
  return r;
}
// interpolation + time blend + interpT of one level from fetched corners (same operation order as hash_t_level)
__device__ __forceinline__ float pair_eval(const PairCorners& pc, const Cell<2>& c, const TimeCoef& tc, bool hi) {
  float a[4], b[4];
  {
    uint32_t gv[2];
    const float w = corner<2>(c, 0, gv);
    const uint4& e = pc.e[0];
    a[0] = fmix0<false>(e.x, w); a[1] = fmix0<true>(e.x, w); a[2] = fmix0<false>(e.y, w); a[3] = fmix0<true>(e.y, w);
    b[0] = fmix0<false>(e.z, w); b[1] = fmix0<true>(e.z, w); b[2] = fmix0<false>(e.w, w); b[3] = fmix0<true>(e.w, w);
  }
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    uint32_t gv[2];
    const float w = corner<2>(c, k, gv);
    const half_t* h = reinterpret_cast<const half_t*>(&pc.e[k]);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      a[f] = fmix(h[f], w, a[f]);
      b[f] = fmix(h[4 + f], w, b[f]);
    }
  }
  float r = 0.0f;
  if (tc.sp.i1 != tc.sp.i2) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      // both slices' features rounded to fp16 by ONE v_cvt_pk_f16_f32, their blend products taken straight from the packed halfs
      // (fmix0: (float)half * w, rounded once -- the value of w * h2f(f2h(x))): 6 instructions per feature instead of 9
      typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
      typedef float ff2 __attribute__((ext_vector_type(2)));
      const ff2 ab = {a[f], b[f]};
      const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, hh2));
      r += tc.basis[f] * (fmix0<false>(w, tc.sp.w1) + fmix0<true>(w, tc.sp.w2));
    }
  } else {
#pragma unroll
    for (int f = 0; f < 4; ++f) r += tc.basis[f] * h2f(f2h(hi ? b[f] : a[f]));
  }
  return r;
}
// corners of cell c from `tab` (the level's block of one pair table); lanes with reuse = true keep what `pc` already holds
__device__ __forceinline__ void pair_fetch(const uint4* __restrict__ tab, const GridDesc& g, int lvl, const Cell<2>& c, bool reuse, PairCorners& pc) {
  const bool hashed = (g.hashed_mask >> lvl) & 1u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t gv[2];
    (void)corner<2>(c, k, gv);
    if (!reuse) pc.e[k] = tab[grid_index<2>(gv, g.res[lvl], g.size[lvl], hashed)];
  }
}

// one HashGridT level (F = 4): fp16-rounded slice features, fp32 blend, interpT
__device__ __forceinline__ float hash_t_level(const FieldDesc& fd, int plane, int lvl, const TimeCoef& tc, const float xy[2]) {
  const GridDesc& g = fd.hd[plane];
  const size_t off = (size_t)g.offset[lvl] * 4;
  const bool hashed = (g.hashed_mask >> lvl) & 1u;
  float a[4], b[4];
  if (fd.hd_pairs[plane]) {  // both slices of a corner in one 16-byte load
    const PairSel ps = pair_sel(tc.sp, fd.n_slices);
    const uint4* tab = reinterpret_cast<const uint4*>(fd.hd_pairs[plane]) + (size_t)ps.q * fd.hd_entries[plane] + g.offset[lvl];
    Cell<2> c = locate<2>(xy, g.scale[lvl]);
#pragma unroll
    for (int f = 0; f < 4; ++f) a[f] = b[f] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t gv[2];
      const float w = corner<2>(c, k, gv);
      const uint4 raw = tab[grid_index<2>(gv, g.res[lvl], g.size[lvl], hashed)];
      const half_t* h = reinterpret_cast<const half_t*>(&raw);
#pragma unroll
      for (int f = 0; f < 4; ++f) {  // same order and roundings as level_lookup: fmix per corner
        a[f] = fmix(h[f], w, a[f]);
        b[f] = fmix(h[4 + f], w, b[f]);
      }
    }
    float r = 0.0f;
    if (tc.sp.i1 != tc.sp.i2) {
#pragma unroll
      for (int f = 0; f < 4; ++f) r += tc.basis[f] * (tc.sp.w1 * h2f(f2h(a[f])) + tc.sp.w2 * h2f(f2h(b[f])));
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) r += tc.basis[f] * h2f(f2h(ps.hi ? b[f] : a[f]));
    }
    return r;
  }
  level_lookup<2, 4>(fd.hd_tables[plane][tc.sp.i1] + off, g.scale[lvl], g.res[lvl], g.size[lvl], hashed, xy, a);
  float r = 0.0f;
  if (tc.sp.i1 != tc.sp.i2) {
    level_lookup<2, 4>(fd.hd_tables[plane][tc.sp.i2] + off, g.scale[lvl], g.res[lvl], g.size[lvl], hashed, xy, b);
#pragma unroll
    for (int f = 0; f < 4; ++f) r += tc.basis[f] * (tc.sp.w1 * h2f(f2h(a[f])) + tc.sp.w2 * h2f(f2h(b[f])));
  } else {
#pragma unroll
    for (int f = 0; f < 4; ++f) r += tc.basis[f] * h2f(f2h(a[f]));
  }
  return r;
}

// ---- time planes as 1-D rows ------------------------------------------------------------------------------------
// The time coordinate of a frame is the same for every sample of a call, so the two time rows a time-plane tap touches and
// their weights are launch-uniform: rows[s][j][e][x][c] = wy0(e) * plane[y0(e)][x][c] + wy1(e) * plane[y1(e)][x][c] is built
// once per call (35 k floats at the default sizes) and a time-plane sample becomes a 1-D interpolation -- two 32-byte
// texels instead of four.  The kernel is bound by L1 bandwidth on exactly these fp32 texel reads (12 KB per sample), and
// 3/4 of the plane taps belong to time planes (3 planes at x, 3 + 3 at the two warped points).
struct PlaneRows {
  const float* base;  // null: sample the planes directly
  int off[MAX_SCALES][3];
};
#define TROWS_FRAMES 3
static __global__ void __launch_bounds__(256) plane_time_rows_kernel(FieldDesc fd, PlaneRows pr, const float* __restrict__ tinfo, float* __restrict__ rows) {
  constexpr int C = 8;
  const int s = blockIdx.y / 3, j = blockIdx.y % 3, e = blockIdx.z;
  const int W = fd.planes.res[s][j], Ht = fd.planes.res[s][3];
  int y0, y1;
  float wy0, wy1, my;
  axis_tap(tinfo[e], Ht, y0, y1, wy0, wy1, my);
  const float* plane = fd.planes_cl + fd.planes.off[s][j == 0 ? 2 : j == 1 ? 4 : 5];
  float* dst = rows + pr.off[s][j] + e * W * C;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < W * C; i += gridDim.x * blockDim.x)
    dst[i] = plane[(size_t)y0 * W * C + i] * wy0 + plane[(size_t)y1 * W * C + i] * wy1;
}

// product over the three time planes of scale s at frame e, from the 1-D rows
template <int C>
__device__ __forceinline__ void planes_time_group(const FieldDesc& fd, const PlaneRows& pr, int s, int e, const float coord[4], float out[C]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int W = fd.planes.res[s][j];
    int x0, x1;
    float wx0, wx1, mx;
    axis_tap(coord[j], W, x0, x1, wx0, wx1, mx);
    const char* b = reinterpret_cast<const char*>(pr.base + pr.off[s][j] + e * W * C);
    const float4_t* p0 = reinterpret_cast<const float4_t*>(b + (uint32_t)x0 * (C * 4u));
    const float4_t* p1 = reinterpret_cast<const float4_t*>(b + (uint32_t)x1 * (C * 4u));
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      const float4_t a = p0[q], c = p1[q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = a[k] * wx0 + c[k] * wx1;
        out[q * 4 + k] = j == 0 ? v : out[q * 4 + k] * v;
      }
    }
  }
}

// offsets of the row blocks and (optionally) the launch that fills them for the call's three frame times
static inline PlaneRows make_plane_rows(const FieldDesc& d, float* rows) {
  PlaneRows pr;
  pr.base = rows;
  int o = 0;
  for (int s = 0; s < MAX_SCALES; ++s)
    for (int j = 0; j < 3; ++j) {
      pr.off[s][j] = o;
      if (s < d.planes.n_scales) o += TROWS_FRAMES * d.planes.res[s][j] * 8;
    }
  return pr;
}

static inline int make_field(const l4d_field_desc* f, FieldDesc& d) {
  if (f->n_slices > MAX_SLICES || f->n_scales > MAX_SCALES || f->plane_channels != 8 || f->hash_static.n_features != 4 ||
      f->hash_static.n_dims != 3) {
    l4d_set_error(1, "field: needs <= 8 time slices, <= 8 plane scales, 8 plane channels, F = 4 hash features");
    return 1;
  }
  d.hs = make_grid_desc(&f->hash_static);
  d.hs_table = (const half_t*)f->hash_static_table;
  for (int p = 0; p < 3; ++p) {
    if (f->hash_dynamic[p].n_features != 4 || f->hash_dynamic[p].n_dims != 2) {
      l4d_set_error(1, "field: dynamic grids must be 2-D with F = 4");
      return 1;
    }
    d.hd[p] = make_grid_desc(&f->hash_dynamic[p]);
    for (int s = 0; s < MAX_SLICES; ++s) d.hd_tables[p][s] = s < f->n_slices ? (const half_t*)f->hash_dynamic_tables[p][s] : nullptr;
  }
  d.n_slices = f->n_slices;
  for (int p = 0; p < 3; ++p) {
    d.hd_pairs[p] = f->n_slices >= 2 ? (const half_t*)f->hash_dynamic_pairs[p] : nullptr;
    const int L = f->hash_dynamic[p].n_levels;
    d.hd_entries[p] = f->hash_dynamic[p].offset[L - 1] + f->hash_dynamic[p].size[L - 1];
  }
  d.planes.n_scales = f->n_scales;
  for (int s = 0; s < f->n_scales; ++s) {
    for (int k = 0; k < 4; ++k) d.planes.res[s][k] = f->plane_res[s * 4 + k];
    for (int c = 0; c < NPLANES; ++c) d.planes.off[s][c] = f->plane_off[s * NPLANES + c];
  }
  d.planes_cl = f->planes_cl;
  return 0;
}

