// Backward of the fused field evaluation (adjoint of fused.hip:density_encode_fwd_kernel; reference
// model/lidar4d.py:139-179 under autograd) for gfx950.
//
// Why this is not one scatter kernel: on MI355X scattered global fp32 atomics sustain only ~20 G lane-ops/s
// (profiles/r01_ubench_global_atomics.txt) while one training step of 16,384 rays needs ~32 G gradient
// scatter-adds; hex-planes and the 2-D dynamic hash tables are small and hot, so their gradients are accumulated in
// LDS as fixed-point integers (ds_add_u32 / ds_add_u64 run 24x faster than ds_add_f32 and are order-independent,
// i.e. deterministic) and flushed once per workgroup with coalesced atomics.
//
//   0. static 3-D hash grid: sorted scatter (binscatter.hip).
//   1. prep (one thread per sample): reads dX; writes the static planes' per-plane gradient factors
//      gvs[scale][plane][p][8] (product rule already applied; plane-major so that a (scale, plane) pass reads it
//      densely), the dynamic-hash upstream gradient transposed gdynT[col][p], and running maxima for the fixed-point
//      scales.
//   2. planes_dyn (one pass): one LDS row per (scale, time plane, frame) -- the time rows and their weights are the
//      same for every sample, so only the spatial axis is accumulated (int32) and the flush applies the row weights;
//      also the coordinate adjoint of the two warped lookups = d(flow).
//   3. planes_static (passes over <=64 KB row bands of each plane): int32 accumulation from gvs.
//   4. dynhash (passes over (plane, level, entry range)): the 2 slices x 4 features of an entry all receive
//      basis[f] * w_slice * H[entry], so only the scalar H is accumulated (int64), then expanded.
#include "field_dev.h"
#include "wave_dev.h"
#include "binscatter.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#define ST_GVS_MAX 0     // [0..8)  max |gvs| per plane scale
#define ST_GD_MAX 8      // max |dX dynamic-plane columns|
#define ST_VMAX 9        // max |plane parameter| (written by the host side before the launch)
#define ST_DYN_MAX 16    // [16 .. 16 + 3L) max |gdynT| per column
#define ST_SIZE 80

// planes of a group in comb order: static (0,1)(0,2)(1,2) = ci 0,1,3; time (0,3)(1,3)(2,3) = ci 2,4,5
__device__ __forceinline__ constexpr int group_ci(bool time_group, int j) {
  return time_group ? (j == 0 ? 2 : j == 1 ? 4 : 5) : (j == 0 ? 0 : j == 1 ? 1 : 3);
}

template <int C>
__device__ __forceinline__ void group_taps(const FieldDesc& fd, int s, const float coord[4], bool time_group, Tap taps[3],
                                           float v[3][C], int cis[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // j is a compile-time constant after unrolling: taps/v stay in registers
    const int ci = time_group ? (j == 0 ? 2 : j == 1 ? 4 : 5) : (j == 0 ? 0 : j == 1 ? 1 : 3);
    const int a = time_group ? j : (j == 2 ? 1 : 0);
    const int b = time_group ? 3 : (j == 0 ? 1 : 2);
    const int W = fd.planes.res[s][a], H = fd.planes.res[s][b];
    axis_tap(coord[a], W, taps[j].x0, taps[j].x1, taps[j].wx0, taps[j].wx1, taps[j].mx);
    axis_tap(coord[b], H, taps[j].y0, taps[j].y1, taps[j].wy0, taps[j].wy1, taps[j].my);
    sample_plane<C>(fd.planes_cl + fd.planes.off[s][ci], W, taps[j], v[j]);
    cis[j] = ci;
  }
}

// ------------------------------------------------------------------------------------------------
// 1. prep
// ------------------------------------------------------------------------------------------------
// The gradient rows are read through LDS: a thread needs its row in 16-byte pieces spread over the kernel (between plane
// gathers), i.e. 64 different cache lines per load instruction, and the lines did not survive in L1 from one piece to the
// next (12.2 GB fetched for 3.4 GB of rows, profiles/r01_pmc_FETCH_SIZE_c3_v10.txt).  Each wave now copies the two column
// ranges the kernel uses -- plane columns [0, 16 nS) and dynamic-hash columns -- of its 64 rows with full-width coalesced
// loads into LDS (row pitch + 8 halfs: spreads the lanes' rows over the banks) and every piece is served from there.
// Segment bounds: coordinates of the FIRST and LAST sample of every aligned 64-sample segment (= one wavefront's samples in the
// multi-pass kernels below), [axis][segment][2].  planes_static_lds_kernel decides from them whether a wavefront misses a row
// band; read from the [3][P] coordinate arrays, that test touched one 64-byte sector per lane and iteration for 8 bytes
// -- 129 band passes x 196,608 segments x 2 sectors = 3.2 GB, most of what its counters showed above the compulsory bytes
// (9.95 GB fetched for 3.62 GB, profiles/r03_pmc_rd.txt).  4.7 MB here, read densely.
__device__ __forceinline__ void write_seg_bounds(float* __restrict__ segb, int64_t pr, int64_t P, int lane, const float4_t& c4) {
  // lanes of a wave hold consecutive samples starting at a multiple of 64 (callers whose chunks are not whole segments pass
  // segb = nullptr, and a wave that does not start on a segment writes nothing); lanes past P carry the last valid sample
  if (!segb || (lane != 0 && lane != 63) || pr - lane >= P || ((pr - lane) & 63) != 0) return;
  const uint32_t n_seg = (uint32_t)((P + 63) >> 6);  // (3 x n_seg x 8 bytes < 2^32: 3 x 2^26 segments = 1.3e10 samples)
  // scalar base + a 32-bit byte offset per lane (round 5): as three 64-bit per-lane addresses the loop-invariant part was hoisted out
  // of the caller's sample loop into register pairs that did not fit -- one of them lived in scratch and came back behind an
  // s_waitcnt vmcnt(0) inside the loop (12 bytes of scratch in planes_dyn_lds_kernel<true, true>, the only default-path kernel with any)
  typedef __attribute__((address_space(1))) char GlobalByte;
  typedef __attribute__((address_space(1))) float GlobalF32;
  const uint64_t b = reinterpret_cast<uint64_t>(segb);
  GlobalByte* base = (GlobalByte*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
  uint32_t off = ((uint32_t)(pr >> 6) * 2u + (lane == 63 ? 1u : 0u)) * 4u;
  asm volatile("" : "+v"(off));  // (opaque per call)
#pragma unroll
  for (int a = 0; a < 3; ++a) *(GlobalF32*)(base + (off + (uint32_t)a * n_seg * 8u)) = c4[a];
}

#define PREP_THREADS 128
__global__ void __launch_bounds__(PREP_THREADS) field_bwd_prep_kernel(FieldDesc fd, const float* __restrict__ xt,
                                                            const float* __restrict__ tinfo,
                                                            int64_t P, const half_t* __restrict__ dX, int in_pad, float pscale,
                                                            half_t* __restrict__ gvs, half_t* __restrict__ gdynT,
                                                            float* __restrict__ stats, int staged, float* __restrict__ xsoa,
                                                            float* __restrict__ segb) {
  constexpr int C = 8;
  extern __shared__ __attribute__((aligned(16))) half_t prep_lds[];
  const int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = pr < P;
  const int64_t p = valid ? pr : P - 1;  // every lane runs the whole body: the wave helpers need all 64 lanes
  const int lane = __lane_id();
  const float4_t c4 = *reinterpret_cast<const float4_t*>(xt + p * 4);
  const float t0 = tinfo[0];
  const bool has_fwd = tinfo[3] != 0.0f, has_bwd = tinfo[4] != 0.0f;
  const float x0[4] = {c4[0], c4[1], c4[2], t0};
  if (valid) {  // coordinates once more as [3][P]: the multi-pass kernels below read one or two of them per sample and pass,
#pragma unroll  // densely, instead of 16-byte xt rows
    for (int a = 0; a < 3; ++a) xsoa[(int64_t)a * P + pr] = c4[a];
  }
  write_seg_bounds(segb, pr, P, lane, c4);
  const int nS = fd.planes.n_scales;
  const half_t* row = dX + p * in_pad;
  int dyn_shift = 0;  // staged: the dynamic-hash columns sit right behind the plane columns in the LDS row
  if (staged) {
    const int colsA = 2 * nS * C, colD = colsA + fd.hs.n_levels * 4;
    const int L3 = fd.hd[0].n_levels + fd.hd[1].n_levels + fd.hd[2].n_levels;
    const int pitch = colsA + L3 + 8;
    const int wave = threadIdx.x >> 6;
    const int64_t wave_p0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
    half_t* wl = prep_lds + wave * 64 * pitch;
    const int chA = colsA / 8, chB = L3 / 8;
    for (int i = lane; i < 64 * (chA + chB); i += 64) {  // consecutive lanes: consecutive 16-byte pieces of a row
      const int r = i / (chA + chB), c = i - r * (chA + chB);
      const int64_t gp = min(wave_p0 + r, P - 1);
      const int src_col = c < chA ? c * 8 : colD + (c - chA) * 8;
      *reinterpret_cast<uint4*>(wl + r * pitch + c * 8) = *reinterpret_cast<const uint4*>(dX + gp * in_pad + src_col);
    }
    __syncthreads();
    row = wl + lane * pitch;
    dyn_shift = colD - colsA;
  }
  const float c0 = 0.5f + (has_fwd ? 0.0f : 0.25f) + (has_bwd ? 0.0f : 0.25f);
  float gd_max = 0.0f;
  float my_stat = 0.0f;  // lane i collects the wave maximum destined for stats[i] (ST_* indices are all < 64)
  static_assert(ST_DYN_MAX + 3 * L4D_MAX_LEVELS <= 64, "one lane per statistic");

  // ---- hex-planes: static planes' product-rule factors, and the range of the time-plane upstream gradient ----
  for (int s = 0; s < nS; ++s) {
    float gs[C], gd[C];
    {
      uint4 u = *reinterpret_cast<const uint4*>(row + s * C);
      const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
      for (int k = 0; k < C; ++k) gs[k] = valid ? h2f(h[k]) : 0.0f;
      u = *reinterpret_cast<const uint4*>(row + (nS + s) * C);
      h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
      for (int k = 0; k < C; ++k) {
        gd[k] = valid ? h2f(h[k]) : 0.0f;
        gd_max = amax_nf(gd_max, gd[k]);
      }
    }
    Tap taps[3];
    float v[3][C];
    int cis[3];
    group_taps<C>(fd, s, x0, false, taps, v, cis);
    float smax = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      half_t hv[C];
#pragma unroll
      for (int k = 0; k < C; k += 2) {  // packed fp32 products, two channels per instruction
        const float2_t g2 = {gs[k], gs[k + 1]}, va = {v[(j + 1) % 3][k], v[(j + 1) % 3][k + 1]};
        const float2_t vb = {v[(j + 2) % 3][k], v[(j + 2) % 3][k + 1]};
        const float2_t gv = g2 * va * vb;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          hv[k + u] = f2h_grad(gv[u]);
          smax = amax_nf(smax, h2f(hv[k + u]));
        }
      }
      if (valid) *reinterpret_cast<uint4*>(gvs + ((int64_t)(s * 3 + j) * P + p) * C) = *reinterpret_cast<uint4*>(hv);
    }
    smax = wave_max(smax);
    if (lane == ST_GVS_MAX + s) my_stat = smax;
  }
  gd_max = wave_max(gd_max);
  if (lane == ST_GD_MAX) my_stat = gd_max;
  int col = 2 * nS * C;

  col += fd.hs.n_levels * 4;  // static 3-D hash columns: handled by the sorted scatter (binscatter.hip)

  // ---- dynamic hash: transposed upstream gradient (current frame only; neighbours are no_grad) ----
  {
    int cidx = 0;
#pragma unroll
    for (int plane = 0; plane < 3; ++plane) {
      const int L = fd.hd[plane].n_levels;
      for (int lvl = 0; lvl < L; ++lvl, ++cidx) {
        const half_t hv = f2h_grad(h2f(row[col - dyn_shift + lvl]) * c0);
        const float a = valid ? amax_nf(0.0f, h2f(hv)) : 0.0f;
        if (valid) gdynT[(int64_t)cidx * P + p] = hv;
        const float m = wave_max(a);
        if (lane == ST_DYN_MAX + cidx) my_stat = m;
      }
      col += L;
    }
  }
  // one guarded atomic-max per statistic, all of them in flight together (lane i owns stats[i]); issued one by one as
  // they were computed, each was a dependent L2 round trip per wave
  if (my_stat > 0.0f) atomic_max_nonneg(stats + lane, my_stat);
}

// ------------------------------------------------------------------------------------------------
// 2. time planes: all scales, all three frames, one pass
// ------------------------------------------------------------------------------------------------
// The time coordinate of a frame is the same for every sample of the launch, so the two time rows a tap touches and
// their weights are launch-uniform: LDS accumulates, per (scale, plane, frame e), ONE row  S[x][c] = sum gv * w_x  over
// the spatial axis only (int32 fixed point), and the flush distributes it to the rows y0(e), y1(e) of the gradient
// plane with the weights wy0(e), wy1(e).  [scale][plane][frame][W][C] ints = 138 KB at the default configuration.
// ROWS: the time-plane VALUES the product rule needs come from the per-call 1-D rows (field_dev.h PlaneRows: two texels per
// plane instead of four taps), and the coordinate adjoint of a warped lookup is sum_c gv_c (row[x1] - row[x0])_c from the
// same two texels -- 144 instead of 480 texel loads per sample in a kernel that is bound by their latency.
#define TFRAMES 3
// -DFB_PHASE_CLOCK (tools/build_abl.sh, tools/phase_probe.py): cycles (s_memtime) the time-plane kernel's wavefronts spend in every part of a
// sample iteration, sampled (wavefront 0 of every 8th workgroup; every wavefront reading the clock slows a kernel tenfold: binscatter.hip)
#ifdef FB_PHASE_CLOCK
__device__ unsigned long long fb_phase_clk[16];
__global__ void fb_phase_clk_read_kernel(unsigned long long* __restrict__ out, int reset) {  // (into DEVICE memory of the caller: no memcpy calls in the library)
  const int i = threadIdx.x;
  if (i >= 16) return;
  if (out) out[i] = fb_phase_clk[i];
  if (reset) fb_phase_clk[i] = 0ull;
}
extern "C" int l4d_debug_fb_phase_clk(unsigned long long* out_dev, int reset, void* stream) {
  fb_phase_clk_read_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_dev, reset);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
#define FB_CLK_DECL const bool clk_on = (blockIdx.x & 7) == 0 && threadIdx.x < 64; uint32_t clk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint32_t clk_last = clk_on ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
#define FB_CLK(i) if (clk_on) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); clk_acc[i] += now_ - clk_last; clk_last = now_; }
#define FB_CLK_FLUSH if (clk_on && threadIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&fb_phase_clk[i_], (unsigned long long)clk_acc[i_]); atomicAdd(&fb_phase_clk[15], 1ull); }
#else
#define FB_CLK_DECL
#define FB_CLK(i)
#define FB_CLK_FLUSH
#endif
// lanes a run of equal texels is merged over before the LDS atomics (16 = a whole DPP row: 4 scan steps per value; 8: 3 steps)
#define PDYN_MERGE 16
#ifndef LDS_PASS_CHUNK_MIN
#define LDS_PASS_CHUNK_MIN 32768  // the multi-pass LDS kernels take at most P / this many chunks: every workgroup zeroes and flushes its whole window (8192 until round 6: at 1,024 rays static planes 0.20 -> 0.17 ms, dynamic hash 0.16 -> 0.11; C2 0.51 -> 0.49 / 0.62 -> 0.58; C3 has 128 chunks either way)
#endif
#define PSTAT_MERGE 16
#define PDYN_THREADS 768  // 12 waves on the one workgroup a CU can hold (138 KB of LDS; 155 VGPRs allow 3 per SIMD): 3.00 -> 2.73 ms against 512
// PREP: the kernel also does the prep kernel's work for its samples -- static planes' product-rule factors gvs, the transposed
// dynamic-hash gradient gdynT, the SoA coordinates, the statistics of both -- while the sample's dX row and coordinates are in
// registers anyway: one pass over dX instead of two, one launch less, and the plane gathers of that part (texel-bandwidth-bound)
// overlap with the segmented scans of this one (VALU-bound).  Needs stats[ST_GD_MAX] from elsewhere: the sigma network's
// backward reports the largest |dX| of the time-plane columns as it stores them (l4d_mlp_bwd dx_absmax).
struct PrepOut {
  half_t* gvs;
  half_t* gdynT;
  float* xsoa;
  float* stats;
  float* segb;
};
// DYN_BATCHES: (PREP) batches of three 16-byte pieces that hold the row's dynamic-hash columns: 1 = up to 24 columns (the default model:
// 3 x 8 levels), 2 = up to 48 (16 levels per stack: the C2-shaped model) -- a compile-time count so that the default kernel stays what it is
template <bool ROWS, bool PREP, int DYN_BATCHES>
__device__ __forceinline__ void planes_dyn_lds_body(FieldDesc fd, float* __restrict__ garena, const float* __restrict__ xt,
                                                    const half_t* __restrict__ flow16, const float* __restrict__ tinfo,
                                                    int64_t P, int64_t chunk, const half_t* __restrict__ dX,
                                                    int in_pad, float pscale, const float* __restrict__ stats,
                                                    half_t* __restrict__ dflow16, PlaneRows prows, PrepOut po) {
  // value arithmetic of THIS function body may contract a * b + c into one fma (texel interpolation, the coordinate adjoint's dot
  // products): gradient values move in their last bit; cell / texel indices come from axis_tap (planes_dev.h), which is
  // compiled under the file-wide -ffp-contract=off and still rounds like the forward pass
#pragma clang fp contract(fast)
  constexpr int C = 8;
  extern __shared__ int lds_i[];
  const int nS = fd.planes.n_scales;
  const float t0 = tinfo[0], t1 = tinfo[1], t2 = tinfo[2];
  const bool has_fwd = tinfo[3] != 0.0f, has_bwd = tinfo[4] != 0.0f;
  const float c0 = 0.5f + (has_fwd ? 0.0f : 0.25f) + (has_bwd ? 0.0f : 0.25f);
  __shared__ int lds_off_s[MAX_SCALES * 3], total_s;
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int s = 0; s < nS; ++s)
      for (int j = 0; j < 3; ++j) {
        lds_off_s[s * 3 + j] = tot;
        tot += TFRAMES * fd.planes.res[s][j] * C;  // time plane j pairs spatial axis j with t
      }
    total_s = tot;
  }
  __syncthreads();
  const int total = total_s;
#define lds_off(s, j) lds_off_s[(s) * 3 + (j)]
  for (int i = threadIdx.x; i < total; i += blockDim.x) lds_i[i] = 0;
  __syncthreads();
  const float vmax = stats[ST_VMAX];
  const float fxs = fx_scale((float)chunk * stats[ST_GD_MAX] * vmax * vmax * 1.01f + 1e-30f, 30);

  // consecutive lanes = consecutive samples of a ray: the plane gathers stay coherent, and the equal-texel runs this
  // creates are merged in registers (row_runs / row_scan) before they reach the LDS atomics
  const int64_t lo_p = (int64_t)blockIdx.x * chunk, hi_p = min(P, lo_p + chunk);
  const int64_t n_iter = (chunk + blockDim.x - 1) / blockDim.x;
  float prep_stat = 0.0f;  // PREP: lane i collects the maximum destined for stats[i] over the whole chunk (one atomic at the end)
  half2_t dyn_max = {(half_t)0.0f, (half_t)0.0f};  // PREP: running max |gdynT| over ALL columns (this lane's samples)
  uint32_t dyn_bad = 0u;   // PREP: sticky "saw inf / nan in a dynamic-hash column"

  typedef uint32_t U4 __attribute__((ext_vector_type(4)));
  FB_CLK_DECL
  for (int64_t it = 0; it < n_iter; ++it) {
    FB_CLK(0)  // loop tail / head
    const int64_t pr = lo_p + it * blockDim.x + threadIdx.x;
    const bool active = pr < hi_p;
    const int64_t p = active ? pr : hi_p - 1;
    const float4_t c4 = *reinterpret_cast<const float4_t*>(xt + p * 4);
    float fl[8];
    {
      uint4 u = *reinterpret_cast<const uint4*>(flow16 + p * 16);
      const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
      for (int k = 0; k < 8; ++k) fl[k] = h2f(h[k]);
    }
    const half_t* row = dX + p * in_pad;
    // pieces 0 .. nS-1: the static planes' gradient columns, nS .. 2 nS - 1: the time planes' (contiguous in the row)
    // (Plain loads on purpose: the row's 16-byte pieces are fetched one per scale and count on the line staying in L1 / L2 in
    // between; marked non-temporal -- tried in round 5 together with the kernel's other streams -- every piece came over the fabric
    // again: 3.61 -> 4.37 ms, gpurun_out/s8.)
    U4 piece_next;
    __builtin_memcpy(&piece_next, row + (PREP ? 0 : nS * C), 16);
    auto next_piece = [&](int k) -> U4 {  // hands out piece k (requested one scale ago) and requests piece k + 1
      U4 cur = piece_next;
      asm volatile("" : "+v"(cur));  // (the wait for piece k stands here, in front of the request for piece k + 1)
      __builtin_memcpy(&piece_next, row + min(k + 1, 2 * nS - 1) * C, 16);
      return cur;
    };
#ifdef FB_PHASE_CLOCK
    asm volatile("" : "+v"(fl[0]), "+v"(fl[5]));
    { float cx = c4[0]; asm volatile("" : "+v"(cx)); }
#endif
    FB_CLK(1)  // coordinates and flow arrived
    if (PREP) {
      const int lane = __lane_id();
      if (active) {
        int64_t pr_o = pr;
        asm volatile("" : "+v"(pr_o));  // (opaque per iteration: hoisted out of the loop the three 64-bit addresses did not fit and one was reloaded from scratch here, behind a wait that also drained the stores in front of it)
#pragma unroll
        for (int a = 0; a < 3; ++a) po.xsoa[(int64_t)a * P + pr_o] = c4[a];
      }
      if (pr - lane < hi_p) write_seg_bounds(po.segb, pr, P, lane, c4);  // (lanes past the chunk carry its last sample)
      const float xs0[4] = {c4[0], c4[1], c4[2], t0};
      for (int s = 0; s < nS; ++s) {  // static planes: gradient of plane j = dX_s * (product of the other two planes' values)
        float gs[C];
        {
          const U4 u = next_piece(s);
          const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
          for (int k = 0; k < C; ++k) gs[k] = active ? h2f(h[k]) : 0.0f;
        }
        Tap taps[3];
        float v[3][C];
        int cis[3];
        group_taps<C>(fd, s, xs0, false, taps, v, cis);
        float smax = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          half_t hv[C];
#pragma unroll
          for (int k = 0; k < C; k += 2) {
            const float2_t g2 = {gs[k], gs[k + 1]}, va = {v[(j + 1) % 3][k], v[(j + 1) % 3][k + 1]};
            const float2_t vb = {v[(j + 2) % 3][k], v[(j + 2) % 3][k + 1]};
            const float2_t gv = g2 * va * vb;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              hv[k + u] = f2h_grad(gv[u]);
              smax = amax_nf(smax, h2f(hv[k + u]));
            }
          }
          if (active) *reinterpret_cast<uint4*>(po.gvs + ((int64_t)(s * 3 + j) * P + p) * C) = *reinterpret_cast<uint4*>(hv);
        }
        smax = wave_max(smax);
        if (lane == ST_GVS_MAX + s) prep_stat = fmaxf(prep_stat, smax);
      }
      FB_CLK(2)  // PREP: static planes' product-rule factors (taps, gvs stores)
      // dynamic hash: the current frame's share c0 of the upstream gradient, transposed (neighbour frames are no_grad)
      const int colD = 2 * nS * C + fd.hs.n_levels * 4;
      const int L3 = fd.hd[0].n_levels + fd.hd[1].n_levels + fd.hd[2].n_levels;
      // Packed halfs throughout: the product by c0 (0.5 / 0.75 / 1: one correctly rounded fp16 multiply, the same value as the
      // fp32 product rounded to fp16), the running maximum of |.| (v_pk_max_f16, reduced over the wave ONCE, after the chunk --
      // 24 wave-level maxima per iteration were a quarter of this part's instructions) and a sticky flag for inf / nan (the
      // packed maximum drops nan), which turns the maximum into +inf at the end: the step is skipped whichever level overflowed.
      const half2_t c0h = {(half_t)c0, (half_t)c0};
      // (the three 16-byte pieces are requested together -- a piece behind the last level re-reads piece 0 and is skipped below:
      // requested one by one, each in front of its own use, they were three memory round trips in a row per iteration)
#pragma unroll
      for (int qb = 0; qb < DYN_BATCHES; ++qb) {
      uint4 ud[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) ud[q] = *reinterpret_cast<const uint4*>(row + colD + ((qb * 3 + q) * 8 < L3 ? (qb * 3 + q) * 8 : 0));
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if ((qb * 3 + q) * 8 < L3) {  // uniform
          const uint4 u = ud[q];
          const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const half2_t hv = __builtin_bit_cast(half2_t, w4[k]) * c0h;
            const uint32_t wv = active ? __builtin_bit_cast(uint32_t, hv) : 0u;
            const int c = (qb * 3 + q) * 8 + 2 * k;
            if (active) {
              po.gdynT[(int64_t)c * P + p] = hv[0];
              po.gdynT[(int64_t)(c + 1) * P + p] = hv[1];
            }
            dyn_bad |= ((wv & 0x7C007C00u) + 0x04000400u) & 0x80008000u;
            const half2_t av = __builtin_bit_cast(half2_t, wv & 0x7FFF7FFFu);
            dyn_max = __builtin_elementwise_max(dyn_max, av);
          }
        }
      }
      }  // batches of three pieces
    }
    FB_CLK(3)  // PREP: dynamic-hash columns transposed
    float gflow[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // d/d(x1), d/d(x2): x1 = x + flow[:3], x2 = x + flow[3:]
    for (int s = 0; s < nS; ++s) {
      float gd[C];
      {
        const U4 u = next_piece(nS + s);
        const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
        for (int k = 0; k < C; ++k) gd[k] = active ? h2f(h[k]) : 0.0f;
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        if ((e == 1 && !has_fwd) || (e == 2 && !has_bwd)) continue;
        const float coef = e == 0 ? c0 : 0.25f;
        const float xe[4] = {c4[0] + (e == 1 ? fl[0] : e == 2 ? fl[3] : 0.0f), c4[1] + (e == 1 ? fl[1] : e == 2 ? fl[4] : 0.0f),
                             c4[2] + (e == 1 ? fl[2] : e == 2 ? fl[5] : 0.0f), e == 0 ? t0 : e == 1 ? t1 : t2};
        Tap taps[3];
        float v[3][C];
        float dv[3][C];  // ROWS: row[x1] - row[x0] per channel (d value / d ix)
        (void)dv;
        int cis[3];
        FB_CLK(4)  // time planes: upstream gradient piece, frame set-up
        if (ROWS) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int W = fd.planes.res[s][j];
            axis_tap(xe[j], W, taps[j].x0, taps[j].x1, taps[j].wx0, taps[j].wx1, taps[j].mx);
            const char* b = reinterpret_cast<const char*>(prows.base + prows.off[s][j] + e * W * C);
            const float4_t* p0 = reinterpret_cast<const float4_t*>(b + (uint32_t)taps[j].x0 * (C * 4u));
            const float4_t* p1 = reinterpret_cast<const float4_t*>(b + (uint32_t)taps[j].x1 * (C * 4u));
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
              const float4_t a = p0[q], c = p1[q];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v[j][q * 4 + k] = a[k] * taps[j].wx0 + c[k] * taps[j].wx1;
                dv[j][q * 4 + k] = c[k] - a[k];
              }
            }
          }
        } else {
          group_taps<C>(fd, s, xe, true, taps, v, cis);
        }
#ifdef FB_PHASE_CLOCK
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(v[j][0]), "+v"(v[j][7]));
#endif
        FB_CLK(5)  // time planes: the frame's row texels loaded and interpolated
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const Tap& t = taps[j];
          const int W = fd.planes.res[s][j];
          float gv[C];
#pragma unroll
          for (int k = 0; k < C; ++k) gv[k] = coef * gd[k] * v[(j + 1) % 3][k] * v[(j + 2) % 3][k];
          if (e > 0) {  // coordinate adjoint of the warped lookups (time plane j pairs spatial axis j with t)
            float gix = 0.0f, giy = 0.0f;
            if (ROWS) {
#pragma unroll
              for (int k = 0; k < C; ++k) gix += dv[j][k] * gv[k];
            } else {
              TapVals<C> tv;
              float dummy[C];
              load_taps<C>(fd.planes_cl + fd.planes.off[s][cis[j]], W, t, tv, dummy);  // L1 hits
              coord_grad_from_taps<C>(tv, t, gv, gix, giy);
            }
            gflow[(e - 1) * 3 + j] += gix * t.mx;
          }
          const RowRuns runs = row_runs<PDYN_MERGE>((uint32_t)t.x0);  // lanes of a run share x0, hence x1 too
          int* acc = &lds_i[lds_off(s, j) + e * W * C];
#pragma unroll
          for (int qx = 0; qx < 2; ++qx) {
            const int xq = qx == 0 ? t.x0 : t.x1;
            const float wxf = (qx == 0 ? t.wx0 : t.wx1) * fxs;
            float vals[C];
#pragma unroll
            for (int k = 0; k < C; ++k) vals[k] = gv[k] * wxf;
            row_scan<C, PDYN_MERGE>(runs, vals);
            if (!runs.tail) continue;  // (inactive lanes carry zeros and a valid clamped key: harmless in any run)
            // (For a given k the lanes of an atomic -- run tails at different texels -- share the four banks = k (mod 8).  Rotating
            // the channel slots by the texel index spreads them over all 32 and was measured SLOWER, 4.14 -> 4.40 ms: the rotated
            // offset is two more VALU instructions per atomic in a kernel that is bound by exactly those, while a constant k rides
            // in the instruction's offset field.)
            int* dst = acc + xq * C;
#pragma unroll
            for (int k = 0; k < C; ++k) atomicAdd(dst + k, fx_round(vals[k]));
          }
        }
        FB_CLK(6)  // time planes: product rule, coordinate adjoint, run scans, LDS atomics of a frame's three planes
      }
    }
    if (active) {  // d(flow), in dX's (loss-scaled) domain
      half_t out[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) out[k] = k < 6 ? f2h_grad(gflow[k]) : (half_t)0.0f;
      uint4* dst = reinterpret_cast<uint4*>(dflow16 + p * 16);
      dst[0] = reinterpret_cast<uint4*>(out)[0];
      dst[1] = reinterpret_cast<uint4*>(out)[1];
    }
    FB_CLK(7)  // d(flow) stored
  }
  FB_CLK_FLUSH
  if (PREP) {
    const int lane = __lane_id();
    const int L3 = fd.hd[0].n_levels + fd.hd[1].n_levels + fd.hd[2].n_levels;
    // one maximum for all dynamic-hash columns: with 30 bits per contribution (dynhash_lds_kernel) a level whose gradients are
    // a thousand times smaller than the largest level's still resolves 2^-20 of its own values (their fp16 payload: 2^-11)
    const bool any_bad = __any(dyn_bad != 0u);
    const float dmax = any_bad ? __builtin_inff() : wave_max(fmaxf(h2f(dyn_max[0]), h2f(dyn_max[1])));
    if (lane >= ST_DYN_MAX && lane < ST_DYN_MAX + L3) prep_stat = fmaxf(prep_stat, dmax);
    if (prep_stat > 0.0f) atomic_max_nonneg(po.stats + lane, prep_stat);
  }
  __syncthreads();
  // an upstream gradient that left the fp16 range (inf / nan in dX) must reach the parameter gradients: the step is
  // then skipped and the loss scale lowered (common.h, f2h_grad)
  if (blockIdx.x == 0 && threadIdx.x == 0 && nonfinite(stats[ST_GD_MAX])) garena[fd.planes.off[0][group_ci(true, 0)]] = __builtin_nanf("");
  // flush: row S of (scale, plane, frame e) goes to the time rows y0(e), y1(e) of the gradient plane
  const float inv = pscale / fxs;
  for (int s = 0; s < nS; ++s) {
    const int Ht = fd.planes.res[s][3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if ((e == 1 && !has_fwd) || (e == 2 && !has_bwd)) continue;
      int y0, y1;
      float wy0, wy1, my;
      axis_tap(e == 0 ? t0 : e == 1 ? t1 : t2, Ht, y0, y1, wy0, wy1, my);
      for (int j = 0; j < 3; ++j) {
        const int W = fd.planes.res[s][j];
        float* g = garena + fd.planes.off[s][group_ci(true, j)];
        const int* acc = &lds_i[lds_off(s, j) + e * W * C];
        for (int i = threadIdx.x; i < W * C; i += blockDim.x) {
          const int v = acc[i];
          if (v == 0) continue;
          const float fv = (float)v * inv;
          atomicAdd(g + (size_t)y0 * W * C + i, fv * wy0);
          if (wy1 != 0.0f) atomicAdd(g + (size_t)y1 * W * C + i, fv * wy1);
        }
      }
    }
  }
#undef lds_off
}
template <bool ROWS, bool PREP>
__global__ void __launch_bounds__(PDYN_THREADS) planes_dyn_lds_kernel(FieldDesc fd, float* __restrict__ garena, const float* __restrict__ xt,
                                                            const half_t* __restrict__ flow16, const float* __restrict__ tinfo,
                                                            int64_t P, int64_t chunk, const half_t* __restrict__ dX,
                                                            int in_pad, float pscale, const float* __restrict__ stats,
                                                            half_t* __restrict__ dflow16, PlaneRows prows, PrepOut po) {
  planes_dyn_lds_body<ROWS, PREP, 1>(fd, garena, xt, flow16, tinfo, P, chunk, dX, in_pad, pscale, stats, dflow16, prows, po);
}
// the same with the preparation pass for up to 48 dynamic-hash columns (16 levels per stack)
__global__ void __launch_bounds__(PDYN_THREADS) planes_dyn_lds_wide_kernel(FieldDesc fd, float* __restrict__ garena, const float* __restrict__ xt,
                                                                 const half_t* __restrict__ flow16, const float* __restrict__ tinfo,
                                                                 int64_t P, int64_t chunk, const half_t* __restrict__ dX,
                                                                 int in_pad, float pscale, const float* __restrict__ stats,
                                                                 half_t* __restrict__ dflow16, PlaneRows prows, PrepOut po) {
  planes_dyn_lds_body<true, true, 2>(fd, garena, xt, flow16, tinfo, P, chunk, dX, in_pad, pscale, stats, dflow16, prows, po);
}

// ------------------------------------------------------------------------------------------------
// 3. static planes: passes over row bands
// ------------------------------------------------------------------------------------------------
#define MAX_TASKS 160
struct BandTasks {
  int n;
  short s[MAX_TASKS], j[MAX_TASKS], row0[MAX_TASKS], nrows[MAX_TASKS];
};

// xt here: the [3][P] coordinate arrays the prep kernel wrote (xsoa)
__global__ void __launch_bounds__(1024) planes_static_lds_kernel(FieldDesc fd, BandTasks tasks, float* __restrict__ garena,
                                                               const float* __restrict__ xt, int64_t P, int64_t chunk,
                                                               int wave_skip, const half_t* __restrict__ gvs, float pscale,
                                                               const float* __restrict__ stats, const float* __restrict__ segb) {
  constexpr int C = 8;
  extern __shared__ int lds_i[];
  const int task = blockIdx.y;
  const int s = tasks.s[task], j = tasks.j[task], row0 = tasks.row0[task], nrows = tasks.nrows[task];
  // static plane j of a scale: comb order (0,1)(0,2)(1,2) -> ci 0,1,3
  const int ci = j == 0 ? 0 : j == 1 ? 1 : 3;
  const int a = COMB_A[ci], b = COMB_B[ci];
  const int W = fd.planes.res[s][a], H = fd.planes.res[s][b];
  const int n_el = nrows * W * C;
  for (int i = threadIdx.x; i < n_el; i += blockDim.x) lds_i[i] = 0;
  __syncthreads();
  const float fxs = fx_scale((float)chunk * stats[ST_GVS_MAX + s] * 1.01f + 1e-30f, 30);
  const int64_t lo_p = (int64_t)blockIdx.x * chunk, hi_p = min(P, lo_p + chunk);
  const int64_t n_iter = (chunk + blockDim.x - 1) / blockDim.x;
  const int lane_ = threadIdx.x & 63;
  for (int64_t it0 = 0; it0 < n_iter; it0 += 64) {
    // Band test for 64 iterations at once.  The 64 lanes of a wave are consecutive samples of ONE ray (samples-per-ray
    // % 64 == 0) and the band coordinate is monotone along them, so the first and last sample decide whether the wave
    // misses this row band; lane i tests iteration it0 + i (two loads, all in flight together instead of a dependent
    // load per iteration).
    unsigned long long todo = ~0ull;
    if (wave_skip) {
      const int64_t pw = lo_p + (it0 + lane_) * blockDim.x + (threadIdx.x & ~63);
      bool hit = false;
      if (it0 + lane_ < n_iter && pw < hi_p) {
        int r0a, r1a, r0b, r1b;
        float w0, w1, m;
        float cf, cl;  // band coordinate of the segment's first / last sample
        if (segb) {  // one dense 8-byte load (chunks start at multiples of 64: pw is a segment start)
          const float2_t fl = *reinterpret_cast<const float2_t*>(segb + ((int64_t)b * ((P + 63) >> 6) + (pw >> 6)) * 2);
          cf = fl[0];
          cl = fl[1];
        } else {
          cf = xt[(int64_t)b * P + pw];
          cl = xt[(int64_t)b * P + min(pw + 63, hi_p - 1)];
        }
        axis_tap(cf, H, r0a, r1a, w0, w1, m);
        axis_tap(cl, H, r0b, r1b, w0, w1, m);
        hit = !(max(r1a, r1b) < row0 || min(r0a, r0b) >= row0 + nrows);
      }
      todo = __ballot(hit);
    }
    while (todo) {
    const int bit = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int64_t it = it0 + bit;
    if (it >= n_iter) break;
    const int64_t pr = lo_p + it * blockDim.x + threadIdx.x;
    const bool active = pr < hi_p;
    const int64_t p = active ? pr : hi_p - 1;
    const float ca = xt[(int64_t)a * P + p], cb = xt[(int64_t)b * P + p];
    // (requested together with the coordinates: behind the band test below it was a second memory latency per iteration; the
    // wave-level test above has already sent nearly every wavefront that gets here into the band)
    const uint4 u = *reinterpret_cast<const uint4*>(gvs + ((int64_t)(s * 3 + j) * P + p) * C);  // dense: 16 B per lane, consecutive
    Tap t;
    axis_tap(cb, H, t.y0, t.y1, t.wy0, t.wy1, t.my);
    const bool in0 = active && t.y0 >= row0 && t.y0 < row0 + nrows;
    const bool in1 = active && t.y1 >= row0 && t.y1 < row0 + nrows && t.y1 != t.y0;
    if (!__any(in0 || in1)) continue;  // wave-uniform (continues the while loop)
    axis_tap(ca, W, t.x0, t.x1, t.wx0, t.wx1, t.mx);
    float gv[C];
    {
      const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
      for (int k = 0; k < C; ++k) gv[k] = h2f(h[k]);
    }
    const float wts[4] = {t.wx0 * t.wy0, t.wx1 * t.wy0, t.wx0 * t.wy1, t.wx1 * t.wy1};
    const int ys[4] = {t.y0, t.y0, t.y1, t.y1}, xs_[4] = {t.x0, t.x1, t.x0, t.x1};
    // runs of samples in one cell (same y0, x0 => same four taps and the same in0 / in1); inactive lanes form their own run
    const RowRuns runs = row_runs<PSTAT_MERGE>(active ? (uint32_t)(t.y0 * W + t.x0) : 0xFFFFFFFFu);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool in = q < 2 ? in0 : in1;
      if (!__any(in)) continue;
      float vals[C];
      const float wq = in ? wts[q] * fxs : 0.0f;  // fixed-point scale folded into the tap weight
#pragma unroll
      for (int k = 0; k < C; k += 2) {
        const float2_t r = float2_t{gv[k], gv[k + 1]} * wq;
        vals[k] = r[0];
        vals[k + 1] = r[1];
      }
      row_scan<C, PSTAT_MERGE>(runs, vals);
      if (!(runs.tail && in)) continue;
      int* dst = &lds_i[((ys[q] - row0) * W + xs_[q]) * C];
#pragma unroll
      for (int k = 0; k < C; ++k) atomicAdd(dst + k, fx_round(vals[k]));
    }
    }  // while (todo)
  }
  __syncthreads();
  const float inv = pscale / fxs;
  float* g = garena + fd.planes.off[s][ci] + (size_t)row0 * W * C;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nonfinite(stats[ST_GVS_MAX + s])) g[0] = __builtin_nanf("");  // overflowed upstream gradient
  for (int i = threadIdx.x; i < n_el; i += blockDim.x) {
    const int v = lds_i[i];
    if (v != 0) atomicAdd(g + i, (float)v * inv);
  }
}

// ------------------------------------------------------------------------------------------------
// 4. dynamic hash: scalar H per entry in LDS (int64 fixed point), then expansion into the two slices
// ------------------------------------------------------------------------------------------------
struct HashTasks {
  int n;
  short plane[MAX_TASKS], lvl[MAX_TASKS];
  int lo[MAX_TASKS], cnt[MAX_TASKS];
  int hoff[MAX_TASKS];  // offset of (plane, level) in Hbuf
};

#define DH_UNROLL 4  // samples per lane whose loads are in flight together (1 = the earlier load -> use chain)
// xt here: the [3][P] coordinate arrays the prep kernel wrote (xsoa)
__global__ void __launch_bounds__(1024) dynhash_lds_kernel(FieldDesc fd, HashTasks tasks, const float* __restrict__ xt, int64_t P,
                                                         int64_t chunk, const half_t* __restrict__ gdynT,
                                                         const float* __restrict__ stats, float* __restrict__ Hbuf) {
  extern __shared__ long long lds_l[];
  const int task = blockIdx.y;
  const int plane = tasks.plane[task], lvl = tasks.lvl[task], cnt = tasks.cnt[task];
  // lo >= 0: the task owns entries [lo, lo + cnt).  lo < 0 (PARITY, hashed power-of-two levels of twice the window): the task owns
  // the entries of parity par = -1 - lo, slot = entry >> 1; its part of Hbuf is parity-major: [par * cnt + slot] (dynhash_expand_kernel)
  const bool parity = tasks.lo[task] < 0;
  const uint32_t par = parity ? (uint32_t)(-1 - tasks.lo[task]) : 0u;
  const int lo = parity ? 0 : tasks.lo[task];
  const int hbase = tasks.hoff[task] + (parity ? (int)par * cnt : lo);
  const GridDesc& g = fd.hd[plane];
  int cidx = lvl;
  for (int q = 0; q < plane; ++q) cidx += fd.hd[q].n_levels;
  const float gmax = stats[ST_DYN_MAX + cidx];
  if (!(gmax > 0.0f)) return;  // no gradient reaches this level at all
  if (nonfinite(gmax)) {       // overflowed upstream gradient: hand it on (expanded into both slices' gradients)
    if (blockIdx.x == 0 && threadIdx.x == 0) Hbuf[hbase] = __builtin_nanf("");
    return;
  }
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) lds_l[i] = 0;
  __syncthreads();
  const float fxs = fx_scale(gmax * 1.01f, 30);  // per contribution 30 bits -> one v_cvt_i32_f32, sign-extended (binscatter.hip, pass 2)
  const int ca = plane == 2 ? 1 : 0, cb = plane == 0 ? 1 : 2;
  const float scale = g.scale[lvl];
  const uint32_t res = g.res[lvl], size = g.size[lvl];
  const bool hashed = (g.hashed_mask >> lvl) & 1u;
  const half_t* gcol = gdynT + (int64_t)cidx * P;
  const int64_t lo_p = (int64_t)blockIdx.x * chunk, hi_p = min(P, lo_p + chunk);
  // PAR: on a hashed power-of-two level the entry index is (x ^ y * PRIME1) & mask with PRIME1 odd, so its lowest bit is (x ^ y) & 1:
  // of a cell's four corners exactly TWO have each parity -- (0, 0) and (1, 1), or (1, 0) and (0, 1).  A task that owns one parity
  // of the table therefore evaluates two corners per sample, both of them its own, where a task that owns an index RANGE evaluates
  // all four and drops half of them on average: the xy stack's levels (two windows each) cost 2 x 2 corner evaluations per sample
  // instead of 2 x 4.  Same contributions, same fixed point, integer accumulation: bit-identical gradients.
  auto walk = [&](auto fast_tag, auto par_tag) {  // FAST: hashed level with a power-of-two table (block-uniform; hashgrid_dev.h grid_index_fast)
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr bool PAR = decltype(par_tag)::value;
    // DH_UNROLL samples per lane and iteration, ALL their loads issued before the first use: at the four wavefronts per SIMD that a
    // 1,024-thread workgroup with 128 KB of LDS leaves, a load -> use -> load chain paid one memory latency (~1 us under load) per
    // sample and wavefront -- the kernel ran at twice its issue floor (profiles/r04_floor_table.md).  Integer accumulation: the
    // order of the adds does not matter, the result is bit-identical.
    for (int64_t p0 = lo_p + threadIdx.x; p0 < hi_p; p0 += (int64_t)DH_UNROLL * blockDim.x) {  // hashed 2-D cells: no same-address pile-up
      uint32_t gh[DH_UNROLL];  // (the half's bits)
      float qa[DH_UNROLL], qb[DH_UNROLL];
#pragma unroll
      for (int u = 0; u < DH_UNROLL; ++u) {
        const int64_t p = p0 + (int64_t)u * blockDim.x;
        const int64_t pc = p < hi_p ? p : p0;  // past the end: a valid address, the value is dropped below
        gh[u] = reinterpret_cast<const unsigned short*>(gcol)[pc];
        qa[u] = xt[(int64_t)ca * P + pc];
        qb[u] = xt[(int64_t)cb * P + pc];
      }
      // every value passes through an empty asm before its first use: otherwise the compiler tests the first gradient as soon as
      // it is loaded and sinks that sample's coordinate loads behind the test -- two dependent latencies again
#pragma unroll
      for (int u = 0; u < DH_UNROLL; ++u) asm volatile("" : "+v"(gh[u]), "+v"(qa[u]), "+v"(qb[u]));
#pragma unroll
      for (int u = 0; u < DH_UNROLL; ++u) {
        const float go = p0 + (int64_t)u * blockDim.x < hi_p ? h2f(__builtin_bit_cast(half_t, (unsigned short)gh[u])) : 0.0f;
        if (go == 0.0f) continue;
        const float q[2] = {qa[u], qb[u]};
        Cell<2> c = locate<2>(q, scale);
        if (PAR) {
          const uint32_t sy = (c.cell[0] ^ c.cell[1] ^ par) & 1u;  // corners (0, sy) and (1, 1 - sy) have this task's parity
          const float wx0 = 1.0f - c.frac[0], wx1 = c.frac[0], wy0 = 1.0f - c.frac[1], wy1 = c.frac[1];  // (corner(): 1.0f * wx * wy)
          const float wa = wx0 * (sy ? wy1 : wy0), wb = wx1 * (sy ? wy0 : wy1);
          const uint32_t ga[2] = {c.cell[0], c.cell[1] + sy}, gb[2] = {c.cell[0] + 1u, c.cell[1] + 1u - sy};
          const uint32_t ia = grid_index_fast<2>(ga, size - 1u) >> 1, ib = grid_index_fast<2>(gb, size - 1u) >> 1;
          atomicAdd(reinterpret_cast<unsigned long long*>(&lds_l[ia]), (unsigned long long)(long long)fx_round(go * wa * fxs));
          atomicAdd(reinterpret_cast<unsigned long long*>(&lds_l[ib]), (unsigned long long)(long long)fx_round(go * wb * fxs));
          continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t gv[2];
          const float w = corner<2>(c, k, gv);
          const int idx = (int)(FAST ? grid_index_fast<2>(gv, size - 1u) : grid_index<2>(gv, res, size, hashed)) - lo;
          if (idx >= 0 && idx < cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&lds_l[idx]), (unsigned long long)(long long)fx_round(go * w * fxs));
        }
      }
    }
  };
  if (parity) walk(std::true_type{}, std::true_type{});  // (the host side asks for it on hashed power-of-two levels only)
  else if (hashed && is_pow2(size)) walk(std::true_type{}, std::false_type{});
  else walk(std::false_type{}, std::false_type{});
  __syncthreads();
  const double inv = 1.0 / (double)fxs;
  float* H = Hbuf + hbase;
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
    const long long v = lds_l[i];
    if (v != 0) atomicAdd(H + i, (float)((double)v * inv));
  }
}

// grad[slice i1][entry][f] += w1 * basis[f] * H[entry] * pscale (and i2 with w2): hash_field.py:65-88 adjoint
__global__ void __launch_bounds__(256) dynhash_expand_kernel(FieldDesc fd, FieldGrads fg, const float* __restrict__ tinfo,
                                                            const float* __restrict__ Hbuf, int plane, int hoff0, float pscale,
                                                            uint32_t parity_levels) {
  const GridDesc& g = fd.hd[plane];
  const int lvl = blockIdx.y;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.size[lvl]) return;
  // (levels the LDS kernel walked by entry parity keep their part of Hbuf parity-major: [parity][entry >> 1])
  const uint32_t hi = (parity_levels >> lvl) & 1u ? (i & 1u) * (g.size[lvl] >> 1) + (i >> 1) : i;
  const float h = Hbuf[hoff0 + g.offset[lvl] + hi] * pscale;
  if (h == 0.0f) return;
  const TimeCoef tc = time_coef(tinfo[0], fd.n_slices);
  float4_t* a = reinterpret_cast<float4_t*>(fg.hd_tables[plane][tc.sp.i1] + ((size_t)g.offset[lvl] + i) * 4);
  float4_t va = *a;
#pragma unroll
  for (int f = 0; f < 4; ++f) va[f] += h * tc.basis[f] * tc.sp.w1;
  *a = va;
  if (tc.sp.i1 != tc.sp.i2) {
    float4_t* b = reinterpret_cast<float4_t*>(fg.hd_tables[plane][tc.sp.i2] + ((size_t)g.offset[lvl] + i) * 4);
    float4_t vb = *b;
#pragma unroll
    for (int f = 0; f < 4; ++f) vb[f] += h * tc.basis[f] * tc.sp.w2;
    *b = vb;
  }
}

// ================================================================================================
// C ABI
// ================================================================================================
static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

struct WorkLayout {
  int64_t gvs, gdynT, xsoa, segb, stats, hbuf, bins, total;
  int64_t hbuf_floats;
};
static WorkLayout work_layout(const l4d_field_desc* f, int64_t P) {
  WorkLayout w;
  int64_t L3 = f->hash_dynamic[0].n_levels + f->hash_dynamic[1].n_levels + f->hash_dynamic[2].n_levels;
  int64_t hb = 0;
  for (int p = 0; p < 3; ++p)
    for (int l = 0; l < f->hash_dynamic[p].n_levels; ++l) hb += f->hash_dynamic[p].size[l];
  w.hbuf_floats = hb;
  w.stats = 0;
  w.hbuf = align256(ST_SIZE * 4);
  w.gvs = w.hbuf + align256(hb * 4);
  w.gdynT = w.gvs + align256(P * f->n_scales * 3 * 8 * 2);
  w.xsoa = w.gdynT + align256(L3 * P * 2);
  w.segb = w.xsoa + align256(3 * P * 4);
  w.bins = w.segb + align256(3 * ((P + 63) / 64) * 2 * 4);
  w.total = w.bins + align256(bs_plan(make_grid_desc(&f->hash_static), 3, 4, P).bytes);
  return w;
}

extern "C" int64_t l4d_density_encode_bwd_workspace(const l4d_field_desc* f, int64_t P) { return work_layout(f, P).total; }

extern "C" int l4d_density_encode_bwd(const l4d_field_desc* f, const l4d_field_grads* g, const float* xt, const void* flow16,
                                      const float* tinfo, int64_t P, const void* dX, int32_t in_pad, float param_scale,
                                      const float* plane_abs_max, int32_t samples_per_ray, void* workspace, void* dflow16,
                                      float* plane_rows, const float* gd_absmax, int32_t defer_join, void* stream_) {
  if (P == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  // Independent parts of the adjoint on side streams (l4d_streams_config bit 1): the sorted scatter of the static grid
  // (HBM-streaming + LDS ranking) needs dX only; the static-plane and dynamic-hash adjoints (LDS atomics) need the prep
  // kernel's outputs; the time planes (VALU-bound) stay on the launch stream -- they produce d(flow), which the caller's flow
  // network backward waits for.  The kernels are bound by different units and share the chip instead of queueing.
  const bool forked = (l4d_streams_mask() & 2) && P >= (1 << 18);
  hipStream_t s_bins = stream, s_lds = stream;
  FieldDesc d;
  if (make_field(f, d)) return 1;
  FieldGrads fg;
  fg.hs_table = g->hash_static_table;
  for (int p = 0; p < 3; ++p)
    for (int s = 0; s < MAX_SLICES; ++s) fg.hd_tables[p][s] = s < f->n_slices ? g->hash_dynamic_tables[p][s] : nullptr;
  fg.planes_cl = g->planes_cl;
  const int L3 = d.hd[0].n_levels + d.hd[1].n_levels + d.hd[2].n_levels;
  if (ST_DYN_MAX + L3 > ST_SIZE) { l4d_set_error(1, "l4d_density_encode_bwd: too many dynamic hash levels"); return 1; }
  {  // everything that can be rejected is rejected BEFORE the first side stream is forked (an error return behind a
    // fork left that stream un-joined -- under capture an unjoined fork, in eager mode later work not ordered behind it)
    int lds_t = 0;
    for (int s = 0; s < d.planes.n_scales; ++s)
      for (int j = 0; j < 3; ++j) lds_t += TFRAMES * d.planes.res[s][j] * 8 * 4;
    if (lds_t > 160 * 1024) { l4d_set_error(1, "l4d_density_encode_bwd: time planes exceed LDS"); return 1; }
  }
  // error returns behind a fork join what was forked, so that the launch stream is ordered behind the side streams' work again
  auto fail = [&](int rc) -> int {
    if (forked) { (void)l4d_side_join(stream_, 1); (void)l4d_side_join(stream_, 2); }
    return rc;
  };
  const WorkLayout w = work_layout(f, P);
  char* ws = (char*)workspace;
  float* stats = (float*)(ws + w.stats);
  float* Hbuf = (float*)(ws + w.hbuf);
  half_t* gvs = (half_t*)(ws + w.gvs);
  half_t* gdynT = (half_t*)(ws + w.gdynT);
  float* xsoa = (float*)(ws + w.xsoa);  // [3][P] coordinates, written by the prep kernel
  float* segb = (float*)(ws + w.segb);  // [3][P / 64][2] first / last coordinate per 64-sample segment
  l4d_fill_async(ws, 0u, (int64_t)w.gvs, stream);  // stats + Hbuf
  l4d_copy_words_async(stats + ST_VMAX, plane_abs_max, 1, stream);

  // static 3-D hash grid: sorted scatter of dX[:, 2*nS*8 + lvl*4 ..] (binscatter.hip)
  {
    if (forked) {
      s_bins = (hipStream_t)l4d_side_fork(stream_, 1);
      if (!s_bins) return fail(1);
    }
    const int cols3[3] = {0, 1, 2};
    int rc = bs_scatter(d.hs, 3, 4, xt, P, 4, cols3, (const half_t*)dX, in_pad, 2 * d.planes.n_scales * 8, 1.0f, fg.hs_table,
                        param_scale, ws + w.bins, s_bins);
    if (rc) return fail(rc);
  }
  // With the largest |dX| of the time-plane columns known beforehand (gd_absmax, from the sigma network's backward) the prep
  // kernel's work is done by the time-plane kernel itself (planes_dyn_lds_kernel<.., PREP = true>).
  const int colsA_ = 2 * d.planes.n_scales * 8, colD_ = colsA_ + d.hs.n_levels * 4;
  const bool fused_prep = gd_absmax && plane_rows && colD_ % 8 == 0 && L3 % 8 == 0 && L3 <= 48 && in_pad % 8 == 0;
  const bool prep_side = false;  // (the preparation kernel on the side stream of its consumers: measured no gain in round 4, switch removed)
  if (fused_prep || prep_side) {
    l4d_copy_words_async(stats + ST_GD_MAX, gd_absmax, 1, stream);
  }
  if (prep_side) {
    s_lds = (hipStream_t)l4d_side_fork(stream_, 2);
    if (!s_lds) return fail(1);
  }
  if (!fused_prep) {
    const int staged = (colD_ % 8 == 0 && L3 % 8 == 0) ? 1 : 0;  // 16-byte pieces
    const int lds = staged ? PREP_THREADS * (colsA_ + L3 + 8) * 2 : 0;
    L4D_LAUNCH(field_bwd_prep_kernel, dim3((unsigned)ceil_div64(P, PREP_THREADS)), dim3(PREP_THREADS), lds, prep_side ? s_lds : stream, d, xt, tinfo, P,
               (const half_t*)dX, in_pad, param_scale, gvs, gdynT, stats, staged, xsoa, segb);
  }

  if (forked && !fused_prep && !prep_side) {  // after the prep kernel
    s_lds = (hipStream_t)l4d_side_fork(stream_, 2);
    if (!s_lds) return fail(1);
  }
  // chunking: one chunk per workgroup column; few enough chunks that the flush traffic stays small
  // (one chunk per CU from 2,048 samples per chunk on: at the reference's own batch of 1,024 rays -- 786 k samples -- the former
  // rule of >= 8,192 samples per chunk left 96 workgroups for 256 CUs in the time-plane kernel, whose grid is the chunks)
  constexpr int chunk_min = 2048;
  int n_chunks = (int)std::min<int64_t>(256, std::max<int64_t>(1, ceil_div64(P, chunk_min)));
  const int64_t chunk = ceil_div64(P, n_chunks);
  n_chunks = (int)ceil_div64(P, chunk);
  // consecutive-lane = consecutive-sample-of-one-ray property, needed for the wave-level band skip
  const int wave_skip = samples_per_ray > 0 && samples_per_ray % 64 == 0 && chunk % 64 == 0;
  // the multi-pass LDS kernels may cut the samples into fewer, larger chunks than the time-plane kernel (whose grid IS the chunks):
  // every workgroup flushes its whole LDS window once, so fewer chunks = less flush traffic
  auto chunks_for = [&](int n, int* n_out) -> int64_t {
    n = (int)std::min<int64_t>(n, std::max<int64_t>(1, ceil_div64(P, LDS_PASS_CHUNK_MIN)));
    int64_t c = ceil_div64(ceil_div64(P, n), 64) * 64;  // whole 64-sample segments
    *n_out = (int)ceil_div64(P, c);
    return c;
  };
  int n_chunks_ps = n_chunks, n_chunks_dh = n_chunks;
  // measured at C3 (gpurun_out/r4d): 128 chunks 1.67 / 1.47 ms (static planes / dynamic hash), 256: 1.72 / 1.50, 512: 2.00 / 1.60, 64: 1.74 / 1.53
  const int64_t chunk_ps = wave_skip ? chunks_for(128, &n_chunks_ps) : chunk;
  const int64_t chunk_dh = wave_skip ? chunks_for(128, &n_chunks_dh) : chunk;

  // time planes
  {
    int lds = 0;
    for (int s = 0; s < d.planes.n_scales; ++s)
      for (int j = 0; j < 3; ++j) lds += TFRAMES * d.planes.res[s][j] * 8 * 4;
    if (lds > 160 * 1024) { l4d_set_error(1, "l4d_density_encode_bwd: time planes exceed LDS"); return fail(1); }  // (checked above)
    const PlaneRows pr = make_plane_rows(d, plane_rows);
    const PrepOut po{gvs, gdynT, xsoa, stats, wave_skip ? segb : nullptr};  // (read only under wave_skip; chunks of another size do not start on segments)
    if (plane_rows) {
      L4D_LAUNCH(plane_time_rows_kernel, dim3(2, d.planes.n_scales * 3, TROWS_FRAMES), dim3(256), 0, stream, d, pr, tinfo, plane_rows);
      if (fused_prep && L3 > 24) {
        (void)hipFuncSetAttribute((const void*)planes_dyn_lds_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        L4D_LAUNCH(planes_dyn_lds_wide_kernel, dim3(n_chunks), dim3(PDYN_THREADS), lds, stream, d, fg.planes_cl, xt, (const half_t*)flow16, tinfo,
                   P, chunk, (const half_t*)dX, in_pad, param_scale, stats, (half_t*)dflow16, pr, po);
      } else if (fused_prep) {
        (void)hipFuncSetAttribute((const void*)planes_dyn_lds_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        L4D_LAUNCH((planes_dyn_lds_kernel<true, true>), dim3(n_chunks), dim3(PDYN_THREADS), lds, stream, d, fg.planes_cl, xt, (const half_t*)flow16, tinfo,
                   P, chunk, (const half_t*)dX, in_pad, param_scale, stats, (half_t*)dflow16, pr, po);
      } else {
        (void)hipFuncSetAttribute((const void*)planes_dyn_lds_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        L4D_LAUNCH((planes_dyn_lds_kernel<true, false>), dim3(n_chunks), dim3(PDYN_THREADS), lds, stream, d, fg.planes_cl, xt, (const half_t*)flow16, tinfo,
                   P, chunk, (const half_t*)dX, in_pad, param_scale, stats, (half_t*)dflow16, pr, po);
      }
    } else {
      (void)hipFuncSetAttribute((const void*)planes_dyn_lds_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      L4D_LAUNCH((planes_dyn_lds_kernel<false, false>), dim3(n_chunks), dim3(PDYN_THREADS), lds, stream, d, fg.planes_cl, xt, (const half_t*)flow16, tinfo,
                 P, chunk, (const half_t*)dX, in_pad, param_scale, stats, (half_t*)dflow16, pr, po);
    }
  }
  if (forked && fused_prep) {  // the static-plane / dynamic-hash adjoints need what the fused kernel wrote
    s_lds = (hipStream_t)l4d_side_fork(stream_, 2);
    if (!s_lds) return fail(1);
  }
  // static planes
  {
    BandTasks t;
    t.n = 0;
    int max_lds = 0;
    static const int CA[3] = {0, 0, 1}, CB[3] = {1, 2, 2};
    for (int s = 0; s < d.planes.n_scales; ++s)
      for (int j = 0; j < 3; ++j) {
        const int W = d.planes.res[s][CA[j]], H = d.planes.res[s][CB[j]];
#define PLANES_BAND_KB 64  // two workgroups per CU: measured 2.37 -> 2.07 ms against 128 KB bands (32 KB: 2.80)
        int rows = std::max(1, (PLANES_BAND_KB * 1024) / (W * 8 * 4));
        rows = std::min(rows, H);
        for (int r0 = 0; r0 < H; r0 += rows) {
          if (t.n >= MAX_TASKS) { l4d_set_error(1, "l4d_density_encode_bwd: too many plane bands"); return fail(1); }
          t.s[t.n] = s; t.j[t.n] = j; t.row0[t.n] = r0; t.nrows[t.n] = std::min(rows, H - r0);
          max_lds = std::max(max_lds, t.nrows[t.n] * W * 8 * 4);
          ++t.n;
        }
      }
    (void)hipFuncSetAttribute((const void*)planes_static_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    L4D_LAUNCH(planes_static_lds_kernel, dim3(n_chunks_ps, t.n), dim3(1024), max_lds, s_lds, d, t, fg.planes_cl, xsoa, P, chunk_ps,
                       wave_skip, gvs, param_scale, stats, wave_skip ? segb : nullptr);
  }
  // dynamic hash
  {
    int hoff = 0, hoff_plane[3];
#define DYNHASH_LDS_KB 64  // two workgroups per CU: measured 2.10 -> 1.85 ms against 128 KB parts
    for (int p = 0; p < 3; ++p) {
      hoff_plane[p] = hoff;
      hoff += (int)(d.hd[p].offset[d.hd[p].n_levels - 1] + d.hd[p].size[d.hd[p].n_levels - 1]);
    }
    // Levels larger than one 64 KB window (the xy stack: 2^15 entries = 256 KB of int64 accumulators) are walked in table parts,
    // and every part streams all samples again.  Those levels get a launch of their own with 128 KB parts (one workgroup per CU,
    // half the passes: measured 1.51 -> 1.46 ms against 64 KB parts, gpurun_out/r4f); the levels that fit 64 KB keep two workgroups per CU.
    constexpr int big_kb = 128;
    uint32_t parity_levels[3] = {0u, 0u, 0u};
    static int dh_parity = -1;  // (read once, like the library's other A/B switches)
    if (dh_parity < 0) { const char* e = getenv("L4D_DH_PARITY"); dh_parity = (e && e[0] == '0') ? 0 : 1; }
    for (int group = 0; group < 2; ++group) {  // 0: levels that fit DYNHASH_LDS_KB; 1: larger ones
      const int lds_kb = group == 0 ? DYNHASH_LDS_KB : big_kb;
      const int max_entries = (lds_kb * 1024) / 8, fit = (DYNHASH_LDS_KB * 1024) / 8;
      HashTasks t;
      t.n = 0;
      for (int p = 0; p < 3; ++p)
        for (int l = 0; l < d.hd[p].n_levels; ++l) {
          const int size = (int)d.hd[p].size[l];
          if ((size <= fit) != (group == 0)) continue;
          // a hashed power-of-two level of exactly two windows: one task per entry PARITY (two corners per sample each) instead of
          // one per index range (four each); L4D_DH_PARITY=0: the range form (A/B switch, tests/test_gpu_switches.py)
          if (dh_parity && size == 2 * max_entries && is_pow2((uint32_t)size) && ((d.hd[p].hashed_mask >> l) & 1u)) {
            for (int par = 0; par < 2; ++par) {
              if (t.n >= MAX_TASKS) { l4d_set_error(1, "l4d_density_encode_bwd: too many hash tasks"); return fail(1); }
              t.plane[t.n] = p; t.lvl[t.n] = l; t.lo[t.n] = -1 - par; t.cnt[t.n] = size / 2;
              t.hoff[t.n] = hoff_plane[p] + (int)d.hd[p].offset[l];
              ++t.n;
            }
            parity_levels[p] |= 1u << l;
            continue;
          }
          for (int lo = 0; lo < size; lo += max_entries) {
            if (t.n >= MAX_TASKS) { l4d_set_error(1, "l4d_density_encode_bwd: too many hash tasks"); return fail(1); }
            t.plane[t.n] = p; t.lvl[t.n] = l; t.lo[t.n] = lo; t.cnt[t.n] = std::min(max_entries, size - lo);
            t.hoff[t.n] = hoff_plane[p] + (int)d.hd[p].offset[l];
            ++t.n;
          }
        }
      if (t.n == 0) continue;
      (void)hipFuncSetAttribute((const void*)dynhash_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      L4D_LAUNCH(dynhash_lds_kernel, dim3(n_chunks_dh, t.n), dim3(1024), lds_kb * 1024, s_lds, d, t, xsoa, P, chunk_dh, gdynT, stats, Hbuf);
    }
    for (int p = 0; p < 3; ++p) {
      unsigned max_size = 0;
      for (int l = 0; l < d.hd[p].n_levels; ++l) max_size = std::max(max_size, d.hd[p].size[l]);
      L4D_LAUNCH(dynhash_expand_kernel, dim3((max_size + 255) / 256, d.hd[p].n_levels), dim3(256), 0, s_lds, d, fg, tinfo,
                         Hbuf, p, hoff_plane[p], param_scale, parity_levels[p]);
    }
  }
  L4D_LAUNCH_CHECK("l4d_density_encode_bwd");
  if (forked && !defer_join) {
    int rc = l4d_side_join(stream_, 1);
    if (!rc) rc = l4d_side_join(stream_, 2);
    if (rc) return rc;
  }
  return 0;
}
