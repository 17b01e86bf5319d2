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
      for (int f = 0; f < 4; f += 2) {  // same order and roundings as level_lookup: acc + v * w per corner
        const float2_t ra = float2_t{a[f], a[f + 1]} + float2_t{h2f(h[f]), h2f(h[f + 1])} * w;
        const float2_t rb = float2_t{b[f], b[f + 1]} + float2_t{h2f(h[4 + f]), h2f(h[5 + f])} * w;
        a[f] = ra[0]; a[f + 1] = ra[1];
        b[f] = rb[0]; b[f + 1] = rb[1];
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

