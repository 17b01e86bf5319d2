// Internal (C++) interface of the sorted gradient scatter (binscatter.hip).
#pragma once
#include "common.h"

struct BsCols {
  int c[3];
};

#define BS_LISTS 8  // record lists per (level, bin): one per XCD

// Where a level's records go (kernel argument of both passes; all positions in RECORDS from the start of the respective region).
struct BsLayout {
  uint32_t cap[L4D_MAX_LEVELS];        // records per list; list (level, bin, k) starts at list_base[level] + (bin * BS_LISTS + k) * cap
  uint32_t ovf_cap[L4D_MAX_LEVELS];    // records of the level's overflow list (samples x 2^D: everything the level can produce)
  uint64_t list_base[L4D_MAX_LEVELS];
  uint64_t ovf_base[L4D_MAX_LEVELS];
};

struct BsPlan {
  int shift;      // bin = entry >> shift
  int rec_words;  // dwords per record (overflow records carry one more: the bin)
  int64_t n_wg;   // pass-1 workgroups (512 samples each, all levels)
  // byte offsets into the workspace: per-level gradient maxima; list cursors [level][list][bin], overflow cursors [level] and
  // per-bin overflow counters [level][bin] (one zero-filled control block); the record lists; the overflow lists
  int64_t off_max, off_cur, off_ovf_cur, off_ovf_cnt, ctrl_bytes, off_lists, off_ovf, bytes;
  BsLayout lay;
};

BsPlan bs_plan(const GridDesc& d, int n_dims, int NV, int64_t P);

// out[(offset[lvl] + entry) * NV + j] += out_scale * sum_p w(p, entry) * pre_scale * g[p * g_stride + g_col + lvl * NV + j]
int bs_scatter(const GridDesc& desc, int n_dims, int NV, const float* x, int64_t P, int x_stride, const int* cols, const half_t* g,
               int g_stride, int g_col, float pre_scale, float* out, float out_scale, void* workspace, hipStream_t stream);
