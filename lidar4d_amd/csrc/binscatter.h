// Internal (C++) interface of the sorted gradient scatter (binscatter.hip).
#pragma once
#include "common.h"

struct BsCols {
  int c[3];
};

struct BsPlan {
  int shift;      // bin = entry >> shift
  int rec_words;  // dwords per record
  int64_t n_wg;   // pass-1 workgroups per level; each owns a fixed slot of 256 * 2^D records + 129 bin offsets
  int64_t off_max, off_offs, off_bins, bytes;
};

BsPlan bs_plan(const GridDesc& d, int n_dims, int NV, int64_t P);

// out[(offset[lvl] + entry) * NV + j] += out_scale * sum_p w(p, entry) * pre_scale * g[p * g_stride + g_col + lvl * NV + j]
int bs_scatter(const GridDesc& desc, int n_dims, int NV, const float* x, int64_t P, int x_stride, const int* cols, const half_t* g,
               int g_stride, int g_col, float pre_scale, float* out, float out_scale, void* workspace, hipStream_t stream);
