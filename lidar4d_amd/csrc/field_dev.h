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

// one HashGridT level (F = 4): fp16-rounded slice features, fp32 blend, interpT
__device__ __forceinline__ float hash_t_level(const FieldDesc& fd, int plane, int lvl, const TimeCoef& tc, const float xy[2]) {
  const GridDesc& g = fd.hd[plane];
  const size_t off = (size_t)g.offset[lvl] * 4;
  const bool hashed = (g.hashed_mask >> lvl) & 1u;
  float a[4], b[4];
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
  d.planes.n_scales = f->n_scales;
  for (int s = 0; s < f->n_scales; ++s) {
    for (int k = 0; k < 4; ++k) d.planes.res[s][k] = f->plane_res[s * 4 + k];
    for (int c = 0; c < NPLANES; ++c) d.planes.off[s][c] = f->plane_off[s * NPLANES + c];
  }
  d.planes_cl = f->planes_cl;
  return 0;
}

