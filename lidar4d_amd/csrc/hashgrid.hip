// Multi-resolution hash-grid encoding for gfx950 (tiny-cuda-nn "HashGrid" semantics),
// replacing tcnn.Encoding at the reference call sites model/hash_field.py:47-57,107-117 and
// model/flow_field.py:67-77, plus the fused HashGridT (model/hash_field.py:76-88).
//
// Direct-gather formulation: one thread per (point, level); a block covers 256 consecutive points
// of one level (blockIdx.y = level) so a level's table stays hot in that CU's L1/L2 while
// consecutive samples of a ray (consecutive threads) share cells on the coarse levels.
// Corner entries are fetched as one 4/8/16-byte vector (F = 2/4/8 fp16 features).
#include "common.h"
#include <type_traits>

#include "hashgrid_dev.h"
#include "wave_dev.h"
#include "binscatter.h"

struct Cols {
  int c[3];
};

// ------------------------------------------------------------------------------------------------
// generic forward / backward
// ------------------------------------------------------------------------------------------------
template <int D, int F>
__global__ void __launch_bounds__(256) hashgrid_fwd_kernel(GridDesc desc, const float* __restrict__ x, int64_t P,
                                                          int x_stride, Cols cols, const half_t* __restrict__ table,
                                                          half_t* __restrict__ out, int out_stride) {
  const int lvl = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float xin[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = x[p * x_stride + cols.c[d]];
  float acc[F];
  level_lookup<D, F>(table + (size_t)desc.offset[lvl] * F, desc.scale[lvl], desc.res[lvl], desc.size[lvl],
                     (desc.hashed_mask >> lvl) & 1u, xin, acc);
  half_t h[F];
#pragma unroll
  for (int f = 0; f < F; ++f) h[f] = f2h(acc[f]);
  typename EntryVec<F>::type* dst = reinterpret_cast<typename EntryVec<F>::type*>(out + p * out_stride + lvl * F);
  *dst = *reinterpret_cast<typename EntryVec<F>::type*>(h);
}

// One thread per POINT, all levels: the point's whole output row (L * F halfs) is assembled in LDS and written with
// 16-byte-per-lane coalesced stores, once.  The (point, level) kernel above writes F halfs per thread into rows that
// are completed by other levels' blocks much later: every 64-byte line is written L times in partial pieces, and x is
// re-read once per level -- at 12.6 M points that traffic, not the table gathers, was most of its 4.1 ms.
#define HG_ROWS_THREADS 128
template <int D, int F>
__global__ void __launch_bounds__(HG_ROWS_THREADS) hashgrid_fwd_rows_kernel(GridDesc desc, const float* __restrict__ x, int64_t P,
                                                                          int x_stride, Cols cols, const half_t* __restrict__ table,
                                                                          half_t* __restrict__ out, int out_stride) {
  extern __shared__ __attribute__((aligned(16))) half_t hg_stage[];
  const int width = desc.n_levels * F;       // halfs per row, a multiple of 8
  const int pitch = width + 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wave_p0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
  const int64_t p = min(wave_p0 + lane, P - 1);  // every thread runs the whole body (block barrier below)
  float xin[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = x[p * x_stride + cols.c[d]];
  half_t* row = hg_stage + (wave * 64 + lane) * pitch;
  for (int lvl = 0; lvl < desc.n_levels; ++lvl) {
    float acc[F];
    level_lookup<D, F>(table + (size_t)desc.offset[lvl] * F, desc.scale[lvl], desc.res[lvl], desc.size[lvl],
                       (desc.hashed_mask >> lvl) & 1u, xin, acc);
    half_t h[F];
#pragma unroll
    for (int f = 0; f < F; ++f) h[f] = f2h(acc[f]);
    *reinterpret_cast<typename EntryVec<F>::type*>(row + lvl * F) = *reinterpret_cast<typename EntryVec<F>::type*>(h);
  }
  __syncthreads();  // rows complete before the cooperative copy-out
  const half_t* wst = hg_stage + wave * 64 * pitch;
  const int chunks = width / 8;
  for (int i = lane; i < 64 * chunks; i += 64) {
    const int r = i / chunks, c = i - r * chunks;
    if (wave_p0 + r < P)
      *reinterpret_cast<uint4*>(out + (wave_p0 + r) * out_stride + c * 8) = *reinterpret_cast<const uint4*>(wst + r * pitch + c * 8);
  }
}

// native vector types (the non-temporal builtins do not take HIP's uint2 / uint4 structs)
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int F> struct NtVec;
template <> struct NtVec<2> { typedef uint32_t type; };
template <> struct NtVec<4> { typedef u32x2_t type; };
template <> struct NtVec<8> { typedef u32x4_t type; };

// coordinates of point p for the level-major kernels, non-temporal.  The render path's points are rows [x, y, z, t] of four floats:
// one 16-byte load instead of three 4-byte loads into the same 16 bytes (block-uniform test).  The vector form reads the row's
// FOURTH float as well: callers of the public entry points that pass x_stride == 4 and columns (0, 1, 2) must own 4 P floats
// (include/lidar4d_hip.h, l4d_hashgrid_fwd_ws / l4d_hashgrid_t_fwd_ws) -- with any other stride or column choice only
// the named columns are touched.
template <int D>
__device__ __forceinline__ void load_coords_nt(const float* __restrict__ x, int64_t p, int x_stride, const Cols& cols, float xin[D]) {
  if (D == 3 && x_stride == 4 && cols.c[0] == 0 && cols.c[1] == 1 && cols.c[2] == 2 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const float4_t v = __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(x) + p);
#pragma unroll
    for (int d = 0; d < D; ++d) xin[d] = v[d];
    return;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = __builtin_nontemporal_load(x + p * x_stride + cols.c[d]);
}

// Level-major evaluation into a scratch array lvlT[level][P][F]; a second, streaming kernel (or the fused encode kernel)
// assembles rows from it.  A level's table (4 MB at 2^19 entries x 8 B) is as large as one XCD's L2, and a gather costs one
// 128-byte LINE at whichever boundary it crosses: 264 G lane-loads/s chip-wide = the L2's 34 TB/s when the line is L2-resident,
// 65 G/s = the fabric's 8.3 TB/s when it comes from the Infinity Cache -- for 4, 8 and 16 bytes per lane and for every cache
// policy alike (tools/ubench/gather_policy.hip, profiles/r05_ubench_gather_policy.txt).  With all levels evaluated per sample
// (the row kernels, the fused encode) every L2 sees all tables and the fine levels miss nearly always.  Two block orders:
//   ORDER 0 (rounds 2-4): level l on the workgroups that land on XCD l % 8 (the dispatcher places block b on XCD b % 8): each L2
//     holds ONE table, but the levels run side by side and a fine level then has ONE XCD's gather rate (33 G/s) for all its
//     samples -- the launch lasts as long as the finest level on an eighth of the chip;
//   ORDER 1 (round 5): the levels one after the other, chip-wide: the workgroups in flight at any moment (2,048 of 49,152 per
//     level) all read the SAME table, every L2 holds a copy of it, and every level has the whole chip's gather rate.
// The coordinate stream is read with non-temporal loads (it must not evict the table), results go out as coalesced 2F-byte
// non-temporal stores.  (Placement is a performance assumption only: any block -> XCD map gives the same result.)
template <int D, int F, bool PAIRLD = false>
__global__ void __launch_bounds__(256) hashgrid_fwd_levels_kernel(GridDesc desc, const float* __restrict__ x, int64_t P, int x_stride,
                                                                 Cols cols, const half_t* __restrict__ table, int64_t n_tiles,
                                                                 half_t* __restrict__ lvlT, int order) {
  int lvl;
  int64_t tile;
  if (order == 0) {
    const int xcd = blockIdx.x & 7;
    const int64_t q = blockIdx.x >> 3;
    const int li = (int)(q / n_tiles);
    tile = q - (int64_t)li * n_tiles;
    lvl = xcd + 8 * li;
    if (lvl >= desc.n_levels) return;
  } else {
    lvl = (int)(blockIdx.x / n_tiles);
    tile = blockIdx.x - (int64_t)lvl * n_tiles;
  }
  const int64_t p = tile * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float xin[D];
  load_coords_nt<D>(x, p, x_stride, cols, xin);
  float acc[F];
  level_lookup<D, F, PAIRLD>(table + (size_t)desc.offset[lvl] * F, desc.scale[lvl], desc.res[lvl], desc.size[lvl],
                             (desc.hashed_mask >> lvl) & 1u, xin, acc);
  half_t h[F];
#pragma unroll
  for (int f = 0; f < F; ++f) h[f] = f2h(acc[f]);
  typedef typename NtVec<F>::type V;
  __builtin_nontemporal_store(*reinterpret_cast<V*>(h), reinterpret_cast<V*>(lvlT + ((int64_t)lvl * P + p) * F));
}

// block order of the level-major kernel: L4D_HG_ORDER=0 restores the XCD-pinned order of rounds 2-4 (A/B; default 1)
static int hg_order() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("L4D_HG_ORDER"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}

// x-neighbour pairs in one 16-byte load where aligned (hashgrid_dev.h PAIRLD; F = 4): L4D_HS_PAIRLD=0 switches it off (A/B)
L4D_INTERNAL int l4d_hs_pairld() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("L4D_HS_PAIRLD"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}

// The x-neighbour pair loads are legal where every pair is one aligned 16-byte load: F = 4, a 16-byte aligned table, and levels
// that start on an even entry.  ONE predicate for both users (the level-major pre-pass below and the pair loads inside the
// fused encode, fused.hip HSMODE 2: ADVICE r5).
L4D_INTERNAL int l4d_hs_pair_ok(const GridDesc* g, int n_features, const void* table) {
  bool ok = n_features == 4 && l4d_hs_pairld() && (reinterpret_cast<uintptr_t>(table) & 15) == 0;
  for (int l = 0; l < g->n_levels; ++l) ok = ok && (g->offset[l] & 1u) == 0;
  return ok ? 1 : 0;
}

// lvlT[level][P][F] <- grid (library-internal: also the first stage of l4d_density_encode_fwd's static-grid columns)
L4D_INTERNAL int l4d_hashgrid_levels_launch(const GridDesc* g, int n_dims, int n_features, const float* x, int64_t P, int x_stride,
                                            const int* cols3, const void* table, void* lvlT, void* stream) {
  Cols c;
  for (int d = 0; d < 3; ++d) c.c[d] = d < n_dims ? cols3[d] : 0;
  const int64_t n_tiles = ceil_div64(P, 256);
  const int order = hg_order();
  const int64_t n_blocks = order == 0 ? n_tiles * ((g->n_levels + 7) / 8) * 8 : n_tiles * g->n_levels;
  if (n_blocks > 0x7fffffffLL) { l4d_set_error(1, "hashgrid levels: too many workgroups"); return 1; }
  dim3 grid((unsigned)n_blocks), block(256);
#define CALL(D, F)                                                                                                             \
  L4D_LAUNCH((hashgrid_fwd_levels_kernel<D, F>), grid, block, 0, (hipStream_t)stream, *g, x, P, x_stride, c, (const half_t*)table, \
             n_tiles, (half_t*)lvlT, order);
  const bool pair_ok = l4d_hs_pair_ok(g, n_features, table) != 0;
  if (pair_ok) {
    if (n_dims == 2)
      L4D_LAUNCH((hashgrid_fwd_levels_kernel<2, 4, true>), grid, block, 0, (hipStream_t)stream, *g, x, P, x_stride, c, (const half_t*)table, n_tiles, (half_t*)lvlT, order);
    else
      L4D_LAUNCH((hashgrid_fwd_levels_kernel<3, 4, true>), grid, block, 0, (hipStream_t)stream, *g, x, P, x_stride, c, (const half_t*)table, n_tiles, (half_t*)lvlT, order);
  }
  else if (n_dims == 2 && n_features == 2) { CALL(2, 2) }
  else if (n_dims == 2 && n_features == 4) { CALL(2, 4) }
  else if (n_dims == 2 && n_features == 8) { CALL(2, 8) }
  else if (n_dims == 3 && n_features == 2) { CALL(3, 2) }
  else if (n_dims == 3 && n_features == 4) { CALL(3, 4) }
  else if (n_dims == 3 && n_features == 8) { CALL(3, 8) }
  else { l4d_set_error(1, "hashgrid: unsupported n_dims/n_features"); return 1; }
#undef CALL
  return 0;
}

// rows [P, out_stride] <- level-major lvlT [L][P][F]; one thread per point, row staged in LDS, coalesced 16-byte stores
template <int F>
__global__ void __launch_bounds__(HG_ROWS_THREADS) hashgrid_rows_from_levels_kernel(int n_levels, int64_t P, const half_t* __restrict__ lvlT,
                                                                                  half_t* __restrict__ out, int out_stride) {
  extern __shared__ __attribute__((aligned(16))) half_t hg_stage[];
  typedef typename NtVec<F>::type V;
  const int width = n_levels * F, pitch = width + 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wave_p0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
  const int64_t p = min(wave_p0 + lane, P - 1);
  half_t* row = hg_stage + (wave * 64 + lane) * pitch;
  for (int lvl = 0; lvl < n_levels; ++lvl)
    *reinterpret_cast<V*>(row + lvl * F) = __builtin_nontemporal_load(reinterpret_cast<const V*>(lvlT + ((int64_t)lvl * P + p) * F));
  __syncthreads();
  const half_t* wst = hg_stage + wave * 64 * pitch;
  const int chunks = width / 8;
  for (int i = lane; i < 64 * chunks; i += 64) {
    const int r = i / chunks, c = i - r * chunks;
    if (wave_p0 + r < P)
      *reinterpret_cast<uint4*>(out + (wave_p0 + r) * out_stride + c * 8) = *reinterpret_cast<const uint4*>(wst + r * pitch + c * 8);
  }
}

template <int D, int F, bool HALF_IN>
__global__ void __launch_bounds__(256) hashgrid_bwd_kernel(GridDesc desc, const float* __restrict__ x, int64_t P,
                                                          int x_stride, Cols cols, const void* __restrict__ dout,
                                                          int dout_stride, float grad_scale,
                                                          float* __restrict__ grad_table) {
  const int lvl = blockIdx.y;
  const int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = pr < P;
  const int64_t p = valid ? pr : P - 1;  // all 64 lanes stay alive for the wave-level run reduction
  float g_out[F];
  bool any = false;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    float v = HALF_IN ? h2f(reinterpret_cast<const half_t*>(dout)[p * dout_stride + lvl * F + f])
                      : reinterpret_cast<const float*>(dout)[p * dout_stride + lvl * F + f];
    g_out[f] = valid ? v * grad_scale : 0.0f;
    any |= (g_out[f] != 0.0f);
  }
  if (!__any(any)) return;  // wave-uniform; zero upstream gradient contributes nothing (exact)
  float xin[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = x[p * x_stride + cols.c[d]];
  Cell<D> c = locate<D>(xin, desc.scale[lvl]);
  float* gt = grad_table + (size_t)desc.offset[lvl] * F;
#pragma unroll
  for (int k = 0; k < (1 << D); ++k) {
    uint32_t g[D];
    float w = corner<D>(c, k, g);
    uint32_t idx = grid_index<D>(g, desc.res[lvl], desc.size[lvl], (desc.hashed_mask >> lvl) & 1u);
    float vals[F];
#pragma unroll
    for (int f = 0; f < F; ++f) vals[f] = w * g_out[f];
    if (wave_run_reduce<F>(idx, any, vals)) {
#pragma unroll
      for (int f = 0; f < F; ++f)
        if (vals[f] != 0.0f) atomicAdd(gt + (size_t)idx * F + f, vals[f]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// HashGridT: two time slices, linear blend, cubic-Lagrange interpT over the F=4 features of a level.
// Rounding points follow the oracle: each slice's interpolated feature is rounded to fp16 (it is a
// tcnn.Encoding output in the reference), blend and interpT are fp32.
// ------------------------------------------------------------------------------------------------
#define L4D_MAX_SLICES 16
struct SliceTables {
  const half_t* t[L4D_MAX_SLICES];
};
struct SliceGrads {
  float* t[L4D_MAX_SLICES];
};

// F features of a level are split into 4 chunks of F/4 (num_basis = 4): out[lvl*(F/4) + j] = sum_b basis[b] * feat[b*(F/4) + j]
// (hash_field.py:65-74 with F=4; flow_field.py:102-111 with F=8, a single table and no time blend).
template <int D, int F, bool HALF_OUT>
__global__ void __launch_bounds__(256) hashgrid_t_fwd_kernel(GridDesc desc, const float* __restrict__ x, int64_t P,
                                                            int x_stride, Cols cols, SliceTables tabs, int n_slices,
                                                            const float* __restrict__ t_ptr, void* __restrict__ out,
                                                            int out_stride) {
  constexpr int FO = F / 4;
  const int lvl = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float t = *t_ptr;
  const SlicePair sp = slice_pair(t, n_slices);
  float basis[4];
  lagrange4(t, basis);
  float xin[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = x[p * x_stride + cols.c[d]];
  const size_t off = (size_t)desc.offset[lvl] * F;
  const bool hashed = (desc.hashed_mask >> lvl) & 1u;
  float a[F], b[F];
  level_lookup<D, F>(tabs.t[sp.i1] + off, desc.scale[lvl], desc.res[lvl], desc.size[lvl], hashed, xin, a);
  if (sp.i1 != sp.i2) {
    level_lookup<D, F>(tabs.t[sp.i2] + off, desc.scale[lvl], desc.res[lvl], desc.size[lvl], hashed, xin, b);
#pragma unroll
    for (int f = 0; f < F; ++f) a[f] = sp.w1 * h2f(f2h(a[f])) + sp.w2 * h2f(f2h(b[f]));
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) a[f] = h2f(f2h(a[f]));
  }
#pragma unroll
  for (int j = 0; j < FO; ++j) {
    float r = 0.0f;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) r += basis[bb] * a[bb * FO + j];
    if (HALF_OUT)
      reinterpret_cast<half_t*>(out)[p * out_stride + lvl * FO + j] = f2h(r);
    else
      reinterpret_cast<float*>(out)[p * out_stride + lvl * FO + j] = r;
  }
}

// The same, level-major into a scratch array lvlT[level][P][F / 4] fp16 (rows are assembled by hashgrid_rows_from_levels_kernel):
// one table at a time chip-wide, the coordinate stream and the results non-temporal -- as hashgrid_fwd_levels_kernel, and for the
// same reason: a gather that misses L2 costs the fabric a 128-byte line ON TOP of its slot in the address path, and written as 4
// bytes per (sample, level) into 32-byte rows the output of the row kernel above is eight partial passes over every line.
// The time coefficients of a launch -- slice pair, blend weights, the four Lagrange basis values -- computed ONCE, by one thread, into the
// head of the workspace: evaluated per thread (one sample and level each, nothing to amortise them over) the basis' twelve IEEE
// divisions were 130 of the level kernel's 355 instructions per wavefront.
struct TimeHead {
  float basis[4];
  float w1, w2;
  int i1, i2;
};
#define HG_T_HEAD_BYTES 256
__global__ void hashgrid_t_head_kernel(const float* __restrict__ t_ptr, int n_slices, TimeHead* __restrict__ head) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float t = *t_ptr;
  const SlicePair sp = slice_pair(t, n_slices);
  TimeHead h;
  lagrange4(t, h.basis);
  h.w1 = sp.w1; h.w2 = sp.w2; h.i1 = sp.i1; h.i2 = sp.i2;
  *head = h;
}
template <int D, int F>
__global__ void __launch_bounds__(256) hashgrid_t_fwd_levels_kernel(GridDesc desc, const float* __restrict__ x, int64_t P, int x_stride, Cols cols,
                                                                   SliceTables tabs, const TimeHead* __restrict__ head,
                                                                   int64_t n_tiles, half_t* __restrict__ lvlT) {
  constexpr int FO = F / 4;
  const int lvl = (int)(blockIdx.x / n_tiles);
  const int64_t tile = blockIdx.x - (int64_t)lvl * n_tiles;
  const int64_t p = tile * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const TimeHead th = *head;  // (uniform address: scalar loads)
  SlicePair sp;
  sp.i1 = th.i1; sp.i2 = th.i2; sp.w1 = th.w1; sp.w2 = th.w2;
  const float basis[4] = {th.basis[0], th.basis[1], th.basis[2], th.basis[3]};
  float xin[D];
  load_coords_nt<D>(x, p, x_stride, cols, xin);
  const size_t off = (size_t)desc.offset[lvl] * F;
  const bool hashed = (desc.hashed_mask >> lvl) & 1u;
  float a[F], b[F];
  level_lookup<D, F>(tabs.t[sp.i1] + off, desc.scale[lvl], desc.res[lvl], desc.size[lvl], hashed, xin, a);
  if (sp.i1 != sp.i2) {
    level_lookup<D, F>(tabs.t[sp.i2] + off, desc.scale[lvl], desc.res[lvl], desc.size[lvl], hashed, xin, b);
#pragma unroll
    for (int f = 0; f < F; ++f) a[f] = sp.w1 * h2f(f2h(a[f])) + sp.w2 * h2f(f2h(b[f]));
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) a[f] = h2f(f2h(a[f]));
  }
  half_t h[FO];
#pragma unroll
  for (int j = 0; j < FO; ++j) {
    float r = 0.0f;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) r += basis[bb] * a[bb * FO + j];
    h[j] = f2h(r);
  }
  typedef typename std::conditional<FO == 2, uint32_t, unsigned short>::type V;
  __builtin_nontemporal_store(*reinterpret_cast<V*>(h), reinterpret_cast<V*>(lvlT + ((int64_t)lvl * P + p) * FO));
}

// Adjoint of the fused time blend + interpT.  Every feature of an entry receives basis[f / FO] * w_slice * Hs[f % FO]
// where Hs[j] = sum_p go[p][j] * w_corner: only the FO scalars per entry are scattered (4 x fewer atomics at F = 8,
// 8 x fewer for the two-slice F = 4 case), run-length pre-reduced along the ray, then expanded by a second kernel.
template <int D, int F, bool HALF_IN>
__global__ void __launch_bounds__(256) hashgrid_t_bwd_kernel(GridDesc desc, const float* __restrict__ x, int64_t P,
                                                            int x_stride, Cols cols, const void* __restrict__ dout,
                                                            int dout_stride, float grad_scale, float* __restrict__ Hs) {
  constexpr int FO = F / 4;
  const int lvl = blockIdx.y;
  const int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = pr < P;
  const int64_t p = valid ? pr : P - 1;
  float go[FO];
  bool any = false;
#pragma unroll
  for (int j = 0; j < FO; ++j) {
    const float v = HALF_IN ? h2f(reinterpret_cast<const half_t*>(dout)[p * dout_stride + lvl * FO + j])
                            : reinterpret_cast<const float*>(dout)[p * dout_stride + lvl * FO + j];
    go[j] = valid ? v * grad_scale : 0.0f;
    any |= go[j] != 0.0f;
  }
  if (!__any(any)) return;
  float xin[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xin[d] = x[p * x_stride + cols.c[d]];
  Cell<D> c = locate<D>(xin, desc.scale[lvl]);
  float* h = Hs + (size_t)desc.offset[lvl] * FO;
#pragma unroll
  for (int k = 0; k < (1 << D); ++k) {
    uint32_t g[D];
    const float w = corner<D>(c, k, g);
    const uint32_t idx = grid_index<D>(g, desc.res[lvl], desc.size[lvl], (desc.hashed_mask >> lvl) & 1u);
    float vals[FO];
#pragma unroll
    for (int j = 0; j < FO; ++j) vals[j] = w * go[j];
    if (wave_run_reduce<FO>(idx, any, vals)) {
#pragma unroll
      for (int j = 0; j < FO; ++j)
        if (vals[j] != 0.0f) atomicAdd(h + (size_t)idx * FO + j, vals[j]);
    }
  }
}

template <int F>
__global__ void __launch_bounds__(256) hashgrid_t_expand_kernel(int64_t n_entries, int n_slices, const float* __restrict__ t_ptr,
                                                               const float* __restrict__ Hs, SliceGrads grads) {
  constexpr int FO = F / 4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_entries) return;
  float hs[FO];
  bool any = false;
#pragma unroll
  for (int j = 0; j < FO; ++j) {
    hs[j] = Hs[i * FO + j];
    any |= hs[j] != 0.0f;
  }
  if (!any) return;
  const float t = *t_ptr;
  const SlicePair sp = slice_pair(t, n_slices);
  float basis[4];
  lagrange4(t, basis);
  float* g1 = grads.t[sp.i1] + i * F;
#pragma unroll
  for (int f = 0; f < F; ++f) g1[f] += hs[f % FO] * basis[f / FO] * sp.w1;
  if (sp.i1 != sp.i2) {
    float* g2 = grads.t[sp.i2] + i * F;
#pragma unroll
    for (int f = 0; f < F; ++f) g2[f] += hs[f % FO] * basis[f / FO] * sp.w2;
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static inline Cols make_cols(const int32_t* cols, int D) {
  Cols c;
  for (int d = 0; d < 3; ++d) c.c[d] = d < D ? cols[d] : 0;
  return c;
}

#define DISPATCH_DF(D_, F_, CALL)                        \
  if (D_ == 2 && F_ == 2) { CALL(2, 2) }                 \
  else if (D_ == 2 && F_ == 4) { CALL(2, 4) }            \
  else if (D_ == 2 && F_ == 8) { CALL(2, 8) }            \
  else if (D_ == 3 && F_ == 2) { CALL(3, 2) }            \
  else if (D_ == 3 && F_ == 4) { CALL(3, 4) }            \
  else if (D_ == 3 && F_ == 8) { CALL(3, 8) }            \
  else { l4d_set_error(1, "hashgrid: unsupported n_dims/n_features"); return 1; }

extern "C" int64_t l4d_hashgrid_fwd_workspace(const l4d_grid_desc* desc, int64_t P) {
  return (int64_t)desc->n_levels * P * desc->n_features * 2;
}

extern "C" int l4d_hashgrid_fwd_ws(const l4d_grid_desc* desc, const float* x, int64_t P, int32_t x_stride, const int32_t* cols,
                                   const void* table, void* out, int32_t out_stride, void* workspace, void* stream) {
  if (P == 0) return 0;
  const int width = desc->n_levels * desc->n_features;
  if (!workspace || width % 8 || out_stride % 8 || ((uintptr_t)out & 15) || (desc->n_features != 4 && desc->n_features != 8 && desc->n_features != 2))
    return l4d_hashgrid_fwd(desc, x, P, x_stride, cols, table, out, out_stride, stream);
  GridDesc g = make_grid_desc(desc);
  if (l4d_hashgrid_levels_launch(&g, desc->n_dims, desc->n_features, x, P, x_stride, cols, table, workspace, stream)) return 1;
  const int lds = HG_ROWS_THREADS * (width + 8) * 2;
  dim3 rgrid((unsigned)ceil_div64(P, HG_ROWS_THREADS)), rblock(HG_ROWS_THREADS);
  if (desc->n_features == 2)
    L4D_LAUNCH((hashgrid_rows_from_levels_kernel<2>), rgrid, rblock, lds, (hipStream_t)stream, desc->n_levels, P, (const half_t*)workspace, (half_t*)out, out_stride);
  else if (desc->n_features == 4)
    L4D_LAUNCH((hashgrid_rows_from_levels_kernel<4>), rgrid, rblock, lds, (hipStream_t)stream, desc->n_levels, P, (const half_t*)workspace, (half_t*)out, out_stride);
  else
    L4D_LAUNCH((hashgrid_rows_from_levels_kernel<8>), rgrid, rblock, lds, (hipStream_t)stream, desc->n_levels, P, (const half_t*)workspace, (half_t*)out, out_stride);
  L4D_LAUNCH_CHECK("l4d_hashgrid_fwd_ws");
  return 0;
}

extern "C" int l4d_hashgrid_fwd(const l4d_grid_desc* desc, const float* x, int64_t P, int32_t x_stride,
                                const int32_t* cols, const void* table, void* out, int32_t out_stride,
                                void* stream) {
  if (P == 0) return 0;
  GridDesc g = make_grid_desc(desc);
  Cols c = make_cols(cols, desc->n_dims);
  const int width = desc->n_levels * desc->n_features;
  if (P >= 4096 && width % 8 == 0 && out_stride % 8 == 0 && ((uintptr_t)out & 15) == 0) {  // whole rows, written once
    const int lds = HG_ROWS_THREADS * (width + 8) * 2;
    dim3 rgrid((unsigned)ceil_div64(P, HG_ROWS_THREADS)), rblock(HG_ROWS_THREADS);
#define CALL(D, F)                                                                                                      \
  L4D_LAUNCH((hashgrid_fwd_rows_kernel<D, F>), rgrid, rblock, lds, (hipStream_t)stream, g, x, P, x_stride, c,            \
             (const half_t*)table, (half_t*)out, out_stride);
    DISPATCH_DF(desc->n_dims, desc->n_features, CALL)
#undef CALL
    L4D_LAUNCH_CHECK("l4d_hashgrid_fwd");
    return 0;
  }
  dim3 grid((unsigned)ceil_div64(P, 256), desc->n_levels), block(256);
#define CALL(D, F)                                                                                          \
  L4D_LAUNCH((hashgrid_fwd_kernel<D, F>), grid, block, 0, (hipStream_t)stream, g, x, P, x_stride, c, \
                     (const half_t*)table, (half_t*)out, out_stride);
  DISPATCH_DF(desc->n_dims, desc->n_features, CALL)
#undef CALL
  L4D_LAUNCH_CHECK("l4d_hashgrid_fwd");
  return 0;
}

extern "C" int l4d_hashgrid_bwd(const l4d_grid_desc* desc, const float* x, int64_t P, int32_t x_stride,
                                const int32_t* cols, const void* dout, int32_t dout_stride, int32_t dout_is_half,
                                float grad_scale, float* grad_table, void* stream) {
  if (P == 0) return 0;
  GridDesc g = make_grid_desc(desc);
  Cols c = make_cols(cols, desc->n_dims);
  dim3 grid((unsigned)ceil_div64(P, 256), desc->n_levels), block(256);
#define CALL(D, F)                                                                                                  \
  if (dout_is_half)                                                                                                 \
    L4D_LAUNCH((hashgrid_bwd_kernel<D, F, true>), grid, block, 0, (hipStream_t)stream, g, x, P, x_stride, c, \
                       dout, dout_stride, grad_scale, grad_table);                                                  \
  else                                                                                                              \
    L4D_LAUNCH((hashgrid_bwd_kernel<D, F, false>), grid, block, 0, (hipStream_t)stream, g, x, P, x_stride,  \
                       c, dout, dout_stride, grad_scale, grad_table);
  DISPATCH_DF(desc->n_dims, desc->n_features, CALL)
#undef CALL
  L4D_LAUNCH_CHECK("l4d_hashgrid_bwd");
  return 0;
}

static int check_t(const l4d_grid_desc* desc, int n_slices) {
  if ((desc->n_features != 4 && desc->n_features != 8) || n_slices > L4D_MAX_SLICES || n_slices < 1 ||
      (desc->n_dims != 2 && desc->n_dims != 3)) {
    l4d_set_error(1, "hashgrid_t: needs n_features in {4,8} (num_basis 4), n_dims in {2,3}, 1..16 slices");
    return 1;
  }
  return 0;
}

#define DISPATCH_T(D_, F_, B_, CALL)                                   \
  if (D_ == 2 && F_ == 4 && B_) { CALL(2, 4, true) }                   \
  else if (D_ == 2 && F_ == 4) { CALL(2, 4, false) }                   \
  else if (D_ == 3 && F_ == 4 && B_) { CALL(3, 4, true) }              \
  else if (D_ == 3 && F_ == 4) { CALL(3, 4, false) }                   \
  else if (D_ == 2 && F_ == 8 && B_) { CALL(2, 8, true) }              \
  else if (D_ == 2 && F_ == 8) { CALL(2, 8, false) }                   \
  else if (D_ == 3 && F_ == 8 && B_) { CALL(3, 8, true) }              \
  else { CALL(3, 8, false) }

extern "C" int l4d_hashgrid_t_fwd(const l4d_grid_desc* desc, const float* x, int64_t P, int32_t x_stride,
                                  const int32_t* cols, const void* const* tables, int32_t n_slices, const float* t,
                                  void* out, int32_t out_stride, int32_t out_is_half, void* stream) {
  if (P == 0) return 0;
  if (check_t(desc, n_slices)) return 1;
  GridDesc g = make_grid_desc(desc);
  Cols c = make_cols(cols, desc->n_dims);
  SliceTables tabs;
  for (int i = 0; i < L4D_MAX_SLICES; ++i) tabs.t[i] = i < n_slices ? (const half_t*)tables[i] : nullptr;
  dim3 grid((unsigned)ceil_div64(P, 256), desc->n_levels), block(256);
#define CALL(D, F, B)                                                                                                \
  L4D_LAUNCH((hashgrid_t_fwd_kernel<D, F, B>), grid, block, 0, (hipStream_t)stream, g, x, P, x_stride, c, tabs, \
                     n_slices, t, out, out_stride);
  DISPATCH_T(desc->n_dims, desc->n_features, out_is_half, CALL)
#undef CALL
  L4D_LAUNCH_CHECK("l4d_hashgrid_t_fwd");
  return 0;
}

// level-major form (default from 2^18 points on; L4D_FLOW_LEVELS=0: the row kernel, A/B)
extern "C" int64_t l4d_hashgrid_t_fwd_workspace(const l4d_grid_desc* desc, int64_t P) {
  return HG_T_HEAD_BYTES + (int64_t)desc->n_levels * P * (desc->n_features / 4) * 2;  // time coefficients, then the level-major columns
}

extern "C" int l4d_hashgrid_t_fwd_ws(const l4d_grid_desc* desc, const float* x, int64_t P, int32_t x_stride,
                                     const int32_t* cols, const void* const* tables, int32_t n_slices, const float* t,
                                     void* out, int32_t out_stride, int32_t out_is_half, void* workspace, void* stream) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("L4D_FLOW_LEVELS"); enabled = (e && e[0] == '0') ? 0 : 1; }
  const int FO = desc->n_features / 4, width = desc->n_levels * FO;
  const int64_t n_tiles = ceil_div64(P, 256);
  if (!workspace || !enabled || !out_is_half || P < (1 << 18) || desc->n_features != 8 || desc->n_dims != 3 || width % 8 || out_stride % 8 ||
      ((uintptr_t)out & 15) || n_tiles * desc->n_levels > 0x7fffffffLL)
    return l4d_hashgrid_t_fwd(desc, x, P, x_stride, cols, tables, n_slices, t, out, out_stride, out_is_half, stream);
  if (check_t(desc, n_slices)) return 1;
  GridDesc g = make_grid_desc(desc);
  Cols c = make_cols(cols, desc->n_dims);
  SliceTables tabs;
  for (int i = 0; i < L4D_MAX_SLICES; ++i) tabs.t[i] = i < n_slices ? (const half_t*)tables[i] : nullptr;
  TimeHead* head = (TimeHead*)workspace;
  half_t* lvlT = (half_t*)((char*)workspace + HG_T_HEAD_BYTES);
  L4D_LAUNCH(hashgrid_t_head_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, n_slices, head);
  L4D_LAUNCH((hashgrid_t_fwd_levels_kernel<3, 8>), dim3((unsigned)(n_tiles * desc->n_levels)), dim3(256), 0, (hipStream_t)stream, g, x, P, x_stride,
             c, tabs, (const TimeHead*)head, n_tiles, lvlT);
  const int lds = HG_ROWS_THREADS * (width + 8) * 2;
  L4D_LAUNCH((hashgrid_rows_from_levels_kernel<2>), dim3((unsigned)ceil_div64(P, HG_ROWS_THREADS)), dim3(HG_ROWS_THREADS), lds, (hipStream_t)stream,
             desc->n_levels, P, (const half_t*)lvlT, (half_t*)out, out_stride);
  L4D_LAUNCH_CHECK("l4d_hashgrid_t_fwd_ws");
  return 0;
}

extern "C" int64_t l4d_hashgrid_t_bwd_workspace(const l4d_grid_desc* desc, int64_t P) {
  return bs_plan(make_grid_desc(desc), desc->n_dims, desc->n_features / 4, P).bytes;
}

extern "C" int l4d_hashgrid_t_bwd(const l4d_grid_desc* desc, const float* x, int64_t P, int32_t x_stride,
                                  const int32_t* cols, int32_t n_slices, const float* t, const void* dout,
                                  int32_t dout_stride, int32_t dout_is_half, float grad_scale,
                                  float* const* grad_tables, float* scratch, void* workspace, void* stream) {
  if (P == 0) return 0;
  if (check_t(desc, n_slices)) return 1;
  GridDesc g = make_grid_desc(desc);
  Cols c = make_cols(cols, desc->n_dims);
  SliceGrads gr;
  for (int i = 0; i < L4D_MAX_SLICES; ++i) gr.t[i] = i < n_slices ? grad_tables[i] : nullptr;
  int64_t n_entries = 0;
  for (int l = 0; l < desc->n_levels; ++l) n_entries += desc->size[l];
  const int FO = desc->n_features / 4;
  l4d_fill_async(scratch, 0u, n_entries * FO * (int64_t)sizeof(float), (hipStream_t)stream);
  dim3 block(256);
  if (workspace && dout_is_half) {
    // sorted scatter (binscatter.hip): no scattered global atomics on the hashed levels
    int rc = bs_scatter(g, desc->n_dims, FO, x, P, x_stride, cols, (const half_t*)dout, dout_stride, 0, 1.0f, scratch, grad_scale,
                        workspace, (hipStream_t)stream);
    if (rc) return rc;
  } else {
    dim3 grid((unsigned)ceil_div64(P, 256), desc->n_levels);
#define CALL(D, F, B)                                                                                                \
  L4D_LAUNCH((hashgrid_t_bwd_kernel<D, F, B>), grid, block, 0, (hipStream_t)stream, g, x, P, x_stride, c,    \
                     dout, dout_stride, grad_scale, scratch);
    DISPATCH_T(desc->n_dims, desc->n_features, dout_is_half, CALL)
#undef CALL
  }
  dim3 egrid((unsigned)ceil_div64(n_entries, 256));
  if (desc->n_features == 4)
    L4D_LAUNCH((hashgrid_t_expand_kernel<4>), egrid, block, 0, (hipStream_t)stream, n_entries, n_slices, t, scratch, gr);
  else
    L4D_LAUNCH((hashgrid_t_expand_kernel<8>), egrid, block, 0, (hipStream_t)stream, n_entries, n_slices, t, scratch, gr);
  L4D_LAUNCH_CHECK("l4d_hashgrid_t_bwd");
  return 0;
}
