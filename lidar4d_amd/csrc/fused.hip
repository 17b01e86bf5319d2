// Per-point fused field evaluation of LiDAR4D.density (model/lidar4d.py:139-179) for gfx950:
// hex-planes at (x,t) + time planes at the two flow-warped neighbour-frame points, static 3-D hash grid,
// the three 2-D x time HashGridT stacks at the current and both neighbour frames, the 0.5/0.25/0.25 blends and
// the concat -- written straight into the sigma MLP's fp16 input row [planes_s | planes_d | hash_s | hash_d | 1].
// One thread owns one sample point; no intermediate [P, .] tensor is materialised (the reference creates ~40).
// The backward kernel is the exact adjoint: scatters into plane / hash gradients and returns d(flow).
//
// This is the direct-gather formulation (every table entry is fetched through L1/L2).
#include "field_dev.h"
#include "mlp_dev.h"
#include <algorithm>
#include <type_traits>

// tinfo: [0]=t, [1]=t1, [2]=t2, [3]=has_fwd, [4]=has_bwd, [5]=frame_idx  (model/lidar4d.py:143,157-173)
__global__ void time_setup_kernel(const float* __restrict__ t, int num_frames, float* __restrict__ tinfo) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float tv = *t;
  const int f = (int)(tv * (float)(num_frames - 1));  // int(t * (num_frames - 1)), fp32 product, truncation
  tinfo[0] = tv;
  tinfo[1] = (float)((double)(f + 1) / (double)num_frames);  // torch.tensor((frame_idx + 1) / self.num_frames)
  tinfo[2] = (float)((double)(f - 1) / (double)num_frames);
  tinfo[3] = (f < num_frames - 1) ? 1.0f : 0.0f;
  tinfo[4] = (f > 0) ? 1.0f : 0.0f;
  tinfo[5] = (float)f;
}

template <int C>
__device__ __forceinline__ void planes_group(const FieldDesc& fd, int s, const float coord[4], bool time_group, float out[C]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int ci = time_group ? (j == 0 ? 2 : j == 1 ? 4 : 5) : (j == 0 ? 0 : j == 1 ? 1 : 3);
    const int a = time_group ? j : (j == 2 ? 1 : 0);
    const int b = time_group ? 3 : (j == 0 ? 1 : 2);
    Tap t;
    const int W = fd.planes.res[s][a], H = fd.planes.res[s][b];
    axis_tap(coord[a], W, t.x0, t.x1, t.wx0, t.wx1, t.mx);
    axis_tap(coord[b], H, t.y0, t.y1, t.wy0, t.wy1, t.my);
    float v[C];
    sample_plane<C>(fd.planes_cl + fd.planes.off[s][ci], W, t, v);
#pragma unroll
    for (int k = 0; k < C; ++k) out[k] = j == 0 ? v[k] : out[k] * v[k];
  }
}

__device__ __forceinline__ void store8h(half_t* dst, const float v[8]) {
  half_t h[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) h[k] = f2h(v[k]);
  *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(h);
}

// Each thread assembles its sample's whole input row, but in pieces (16-B plane groups, 8-B hash levels, 2-B dynamic
// levels).  Written straight to HBM those partial-line stores cost 17.7 GB of write traffic for a 3.2 GB matrix
// (profiles/r01_pmc_WRITE_SIZE_c3_v6.txt), so the row is staged in LDS (row pitch in_pad + 8 halfs: 16-B aligned, spreads
// the lanes' rows over the banks) and each wave then writes its 64 rows as full 16-B-per-lane coalesced stores.
#define ENC_THREADS 64  // one wave per workgroup: measured best (64: 13.8 ms, 128: 14.1, 256: 14.5 for the entry point)
#define ENC_MAX_IN_PAD 192  // widest network input row (BASELINE config C2: L = 16 hash levels -> 176 columns)
#define ENC_WAVES_PER_EU 2
#define ENC_SPLIT_MIN_POINTS (1 << 18)  // below this the extra launch and the re-read of xt / flow cost more than the overlap buys
#define ENC_WAVES_PER_EU_HASH 4
// PART: 0 = the whole row in one kernel; 1 = the plane columns [0, 2 nS C) only; 2 = everything behind them (hash grids, ones).
// The split (l4d_density_encode_fwd with side streams) exists for two reasons: the plane part does not need the xz / yz columns
// that dynhash_fwd_lds_kernel produces, so the two run CONCURRENTLY (texel-bandwidth-bound next to VALU / LDS-bound), and the
// hash part alone needs half the registers, i.e. twice the wavefronts to hide its L2-missing gathers behind.
// HSMODE 1: the static grid's columns come from the level-major pre-pass (hashgrid.hip hashgrid_fwd_levels_kernel, hsT[level][P][4]
// fp16) instead of being gathered here: 8 bytes per level and sample, dense, and the kernel's own gathers (xy stack, planes) no
// longer share the L2s with 33 MB of static tables.  HSMODE 2: gathered here, x-neighbour pairs in one 16-byte load where they
// share an aligned pair (hashgrid_dev.h PAIRLD).  HSMODE 0: gathered here, one 8-byte load per corner (rounds 1-4).
// (Measured and removed: the xy stack through a level-major pre-pass of its own as well -- its tables are 512 KB per level, so the
// pre-pass hit L2 always, but took 2.18 ms where the same lookups cost this kernel 1.34: here their address-path time hides behind
// the plane arithmetic; step 31.65 -> 32.58 ms, profiles/r05_experiment_runs.txt session s5.)
// SIGMA (round 5): the density network's forward pass (tcnn FullyFusedMLP 128 -> 64 -> 16, lidar4d.py:83-93,181; trunc_exp on column 0)
// runs as this kernel's epilogue on the rows it has just staged in LDS -- the operand layout, the fragment order and the order of
// the accumulation are mlp_fwd_kernel<8, 1>'s, so y / act / sigma are bit-identical to the two-kernel path -- instead of a separate
// launch that reads the 3.2 GB row matrix back (1.05 ms at 12.6 M samples).  X is still written: the backward pass reads it.
// The whole 128-column row is staged at once (272-byte pitch) next to the 18 weight fragments (18 KB, shared by the workgroup): eight
// wavefronts per workgroup, one workgroup per CU (154 KB of LDS) = the two wavefronts per SIMD the registers allow anyway; the
// wavefronts only ever touch their own rows, so staging and copy-out synchronise per wavefront, not per workgroup.
#ifndef ENC_HS_EARLY
#define ENC_HS_EARLY 1
#endif
#ifndef ENC_HD_EARLY
#define ENC_HD_EARLY 1
#endif
#define ENC_SIGMA_THREADS 512
#define ENC_SIGMA_FRAGS 18
struct SigmaOut {
  const half_t* w;  // W1 [64, 128] | Wo [16, 64] fp16
  half_t* y;        // [P, 16]
  half_t* act;      // [P, 64] or null
  float* sigma;     // [P]
  int persistent;   // 1: gridDim.x workgroups (one per CU) whose wavefronts each walk many 64-sample groups (see the kernel)
};
// -DENC_PHASE_CLOCK (tools/build_abl.sh, tools/phase_probe.py enc): cycles (s_memtime) a wavefront of the fused encode spends in every part of
// a 64-sample group, sampled (wavefront 0 of every 8th workgroup); not compiled into the shipped library
#ifdef ENC_PHASE_CLOCK
__device__ unsigned long long enc_phase_clk[16];
__global__ void enc_phase_clk_read_kernel(unsigned long long* __restrict__ out, int reset) {
  const int i = threadIdx.x;
  if (i >= 16) return;
  if (out) out[i] = enc_phase_clk[i];
  if (reset) enc_phase_clk[i] = 0ull;
}
extern "C" int l4d_debug_enc_phase_clk(unsigned long long* out_dev, int reset, void* stream) {
  enc_phase_clk_read_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_dev, reset);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
#define ENC_CLK_DECL const bool clk_on = (blockIdx.x & 7) == 0 && threadIdx.x < 64; uint32_t clk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint32_t clk_last = clk_on ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u;
#define ENC_CLK(i) if (clk_on) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); clk_acc[i] += now_ - clk_last; clk_last = now_; }
#define ENC_CLK_FLUSH if (clk_on && threadIdx.x == 0) { for (int i_ = 0; i_ < 15; ++i_) atomicAdd(&enc_phase_clk[i_], (unsigned long long)clk_acc[i_]); atomicAdd(&enc_phase_clk[15], 1ull); }
#else
#define ENC_CLK_DECL
#define ENC_CLK(i)
#define ENC_CLK_FLUSH
#endif
template <bool USE_HDT, bool ROWS, int PART = 0, int HSMODE = 0, bool SIGMA = false>
__global__ void __launch_bounds__(SIGMA ? ENC_SIGMA_THREADS : ENC_THREADS) __attribute__((amdgpu_waves_per_eu(PART == 2 ? ENC_WAVES_PER_EU_HASH : ENC_WAVES_PER_EU, 8))) density_encode_fwd_kernel(FieldDesc fd, const float* __restrict__ xt,
                                                                        const half_t* __restrict__ flow16,
                                                                        const float* __restrict__ tinfo, int64_t P,
                                                                        const half_t* __restrict__ hdT,
                                                                        half_t* __restrict__ X, int in_pad, PlaneRows prows,
                                                                        const half_t* __restrict__ hsT, SigmaOut so) {
  constexpr int C = 8;
  static_assert(!SIGMA || PART == 0, "the network epilogue needs the whole row");
  // the row is staged and written out in two parts (planes | everything else) so that the staging buffer is half as
  // large: LDS is what limits this kernel's occupancy (gather latency needs waves in flight).  (SIGMA: the whole row at once.)
  extern __shared__ __attribute__((aligned(16))) half_t stage_all[];  // (SIGMA: 18 weight fragments, then) [threads][pitch]
  half_t* stage = stage_all + (SIGMA ? ENC_SIGMA_FRAGS * 64 * 8 : 0);
  const int colsA = 2 * fd.planes.n_scales * C;
  const int ENC_PITCH = (SIGMA ? in_pad : max(colsA, in_pad - colsA)) + 8;  // halfs per staged row: 16-byte aligned, spreads rows over the banks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t blk_p0 = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x;
  if (!(SIGMA && so.persistent) && blk_p0 >= P) return;  // idle tile of the rounded-up grid (block-uniform)
  if (SIGMA) {  // weight fragments of mlp_fwd_kernel<8, 1>: 16 of the hidden layer (4 neuron tiles x 4 k-steps), 2 of the output layer
    const int i = lane & 15, g = lane >> 4;
    for (int f = wave; f < ENC_SIGMA_FRAGS; f += ENC_SIGMA_THREADS / 64) {
      const h8 v = f < 16 ? build_frag(so.w, HID, 128, 0, perm_row(f >> 2, i), 32 * (f & 3) + 8 * g)
                          : build_frag(so.w + HID * 128, 16, HID, 0, i, 32 * (f - 16) + 8 * g);
      reinterpret_cast<uint4*>(stage_all)[f * 64 + lane] = *reinterpret_cast<const uint4*>(&v);
    }
    __syncthreads();
  }
  // a wavefront reads back only rows its own lanes staged: (SIGMA) wait for the wavefront's LDS writes instead of a workgroup barrier
  auto stage_sync = [&]() {
    if (SIGMA) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    } else {
      __syncthreads();
    }
  };
  // PERSISTENT (SIGMA): the eight wavefronts of a workgroup start together behind the fragment barrier; with one 64-sample group per
  // wavefront they ran their gather phase and their network / store phase in step, and the launch took as long as encode and network
  // one after the other (4.98 ms against 3.94 + 1.05).  Here a wavefront walks its own sequence of groups -- XCD x keeps the x-th
  // contiguous eighth of the samples, as xcd_tile gives it to the one-group form -- with nothing but its own LDS traffic to wait
  // for, so the wavefronts of a CU drift apart and one's stores overlap another's gathers.
  const int64_t n_groups = (P + 63) >> 6, gpx = (n_groups + 7) >> 3;  // groups per XCD
  const int64_t g_step = (int64_t)(gridDim.x >> 3) * (ENC_SIGMA_THREADS / 64);
  int64_t g_in_xcd = (int64_t)(blockIdx.x >> 3) * (ENC_SIGMA_THREADS / 64) + wave;
  const bool persistent = SIGMA && so.persistent;
  ENC_CLK_DECL
  for (;; g_in_xcd += g_step) {
  ENC_CLK(0)  // loop tail / head
  int64_t wave_p0_ = blk_p0 + wave * 64;
  if (persistent) {
    const int64_t grp = (int64_t)(blockIdx.x & 7) * gpx + g_in_xcd;
    if (g_in_xcd >= gpx || grp >= n_groups) break;  // wave-uniform
    wave_p0_ = grp << 6;
  }
  const int64_t wave_p0 = wave_p0_;
  const int64_t pr = wave_p0 + lane;
  const int64_t p = pr < P ? pr : P - 1;
  // (round 5) everything this kernel streams -- coordinates, flow, the pre-passes' columns in; the row, the network's outputs out --
  // is marked non-temporal: 3.8 GB per launch through L2s whose 4 MB the xy stack's slice-pair tables (4 MB) need for themselves
  typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2_nt __attribute__((ext_vector_type(2)));
  const float4_t c4 = __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(xt + p * 4));
  const float t0 = tinfo[0], t1 = tinfo[1], t2 = tinfo[2];
  const bool has_fwd = tinfo[3] != 0.0f, has_bwd = tinfo[4] != 0.0f;
  float fl[8];
  {
    const u32x4_nt u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(flow16 + p * 16));
    const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
    for (int k = 0; k < 8; ++k) fl[k] = h2f(h[k]);
  }
#ifdef ENC_PHASE_CLOCK
  asm volatile("" : "+v"(fl[0]), "+v"(fl[5]));
  { float cx = c4[0]; asm volatile("" : "+v"(cx)); }
#endif
  ENC_CLK(1)  // coordinates and flow arrived
  const float x0[4] = {c4[0], c4[1], c4[2], t0};
  const float x1[4] = {c4[0] + fl[0], c4[1] + fl[1], c4[2] + fl[2], t1};
  const float x2[4] = {c4[0] + fl[3], c4[1] + fl[4], c4[2] + fl[5], t2};
  half_t* row = stage + (wave * 64 + lane) * ENC_PITCH;
  const int nS = fd.planes.n_scales;

  // (HSMODE 1) the static grid's first eight columns are requested HERE, in front of the plane taps, and staged behind them: they
  // depend on nothing but the sample index, and their HBM round trip then runs under the planes' gathers (16 registers)
  typedef uint32_t hs_col_t __attribute__((ext_vector_type(2)));
  hs_col_t hs_early[8];
  if (HSMODE == 1 && ENC_HS_EARLY) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      hs_early[q] = __builtin_nontemporal_load(reinterpret_cast<const hs_col_t*>(hsT) + (int64_t)min(q, fd.hs.n_levels - 1) * P + p);
    asm volatile("" ::: "memory");  // (the requests stay up here)
  }
  // ---- hex-planes (planes_field.py:87-141; blend lidar4d.py:175) ----
  for (int s = 0; PART != 2 && s < nS; ++s) {
    float ps[C], d0[C], d1[C], d2[C];
    planes_group<C>(fd, s, x0, false, ps);
#ifdef ENC_PHASE_CLOCK
    asm volatile("" : "+v"(ps[0]), "+v"(ps[7]));
#endif
    ENC_CLK(8)  // (static planes of a scale: 12 taps)
    if (ROWS) {
      planes_time_group<C>(fd, prows, s, 0, x0, d0);
      if (has_fwd) planes_time_group<C>(fd, prows, s, 1, x1, d1);
      if (has_bwd) planes_time_group<C>(fd, prows, s, 2, x2, d2);
    } else {
      planes_group<C>(fd, s, x0, true, d0);
      if (has_fwd) planes_group<C>(fd, s, x1, true, d1);
      if (has_bwd) planes_group<C>(fd, s, x2, true, d2);
    }
    float pd[C];
#pragma unroll
    for (int k = 0; k < C; ++k) pd[k] = 0.5f * d0[k] + 0.25f * ((has_fwd ? d1[k] : d0[k]) + (has_bwd ? d2[k] : d0[k]));
    store8h(row + s * C, ps);
    store8h(row + (nS + s) * C, pd);
  }
  ENC_CLK(2)  // hex-planes: static + time rows, all scales, staged
  const half_t* wstage = stage + wave * 64 * ENC_PITCH;
  auto copy_out = [&](int c0, int ncols) {  // this wave's 64 staged rows -> X[:, c0 : c0 + ncols], 16 B per lane, coalesced
    const int chunks = ncols / 8;
    // (row, chunk) of a lane's piece advanced by 64 pieces per iteration instead of divided out anew: the run-time division by
    // `chunks` was 20 instructions in each of the 16 iterations of every 64-sample group -- 6 % of the kernel's VALU instructions
    int r = lane / chunks, c = lane - r * chunks;
    const int dr = 64 / chunks, dc = 64 - dr * chunks;  // (wave-uniform)
    for (int idx = lane; idx < 64 * chunks; idx += 64) {
      const int64_t grow = wave_p0 + r;
      if (grow < P)
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4_nt*>(wstage + r * ENC_PITCH + c * 8), reinterpret_cast<u32x4_nt*>(X + grow * in_pad + c0 + c * 8));
      r += dr;
      c += dc;
      if (c >= chunks) {
        c -= chunks;
        ++r;
      }
    }
  };
  if (PART != 2 && !SIGMA) {
    __syncthreads();
    copy_out(0, colsA);
    if (PART == 1) return;
    __syncthreads();
  }
  if (!SIGMA) row -= colsA;  // the second part is staged from column 0 again
  int col = colsA;

  // ---- static 3-D hash grid (hash_field.py:141-144) ----
  {
    const float xs[3] = {x0[0], x0[1], x0[2]};
    constexpr bool HSPRE = HSMODE == 1;
    // Eight levels' columns requested TOGETHER, then staged: written level by level the loop was compiled as two loads, a wait, two
    // LDS stores, ... -- four HBM round trips in a row per sample group (5.6 k of a group's 89 k cycles: tools/phase_probe.py enc)
    for (int l0 = 0; HSPRE && l0 < fd.hs.n_levels; l0 += 8) {
      typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
      u32x2_t v[8];
      if (ENC_HS_EARLY && l0 == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = hs_early[q];
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q)  // (a level behind the last re-reads the last one: unconditional loads, nothing waits in between)
          v[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(hsT) + (int64_t)min(l0 + q, fd.hs.n_levels - 1) * P + p);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(v[q]));
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (l0 + q < fd.hs.n_levels) *reinterpret_cast<u32x2_t*>(row + col + (l0 + q) * 4) = v[q];
    }
    for (int lvl = 0; !HSPRE && lvl < fd.hs.n_levels; ++lvl) {
      float a[4];
      level_lookup<3, 4, HSMODE == 2>(fd.hs_table + (size_t)fd.hs.offset[lvl] * 4, fd.hs.scale[lvl], fd.hs.res[lvl], fd.hs.size[lvl],
                                      (fd.hs.hashed_mask >> lvl) & 1u, xs, a);
      half4_t h;
#pragma unroll
      for (int f = 0; f < 4; ++f) h[f] = f2h(a[f]);
      *reinterpret_cast<half4_t*>(row + col + lvl * 4) = h;
    }
    col += fd.hs.n_levels * 4;
  }

  ENC_CLK(3)  // static grid's columns (pre-pass) loaded and staged
  // ---- dynamic HashGridT stacks at the current and the two warped neighbour frames (lidar4d.py:145,157-176) ----
  const int col_dyn0 = col;
  {
    const TimeCoef tc0 = time_coef(t0, fd.n_slices), tc1 = time_coef(t1, fd.n_slices), tc2 = time_coef(t2, fd.n_slices);
    // (ENC_HD_EARLY) the xz / yz stacks' first eight columns each -- produced by dynhash_fwd_lds_kernel, dependent on nothing but the
    // sample index -- are requested in front of the xy stack's gathers and staged behind them (two columns per register)
    uint32_t hd_early[2][4];
    if (USE_HDT && ENC_HD_EARLY) {
      const int L0 = fd.hd[0].n_levels;
#pragma unroll
      for (int pl = 1; pl < 3; ++pl) {
        const int L = fd.hd[pl].n_levels, cb = L0 + (pl == 2 ? fd.hd[1].n_levels : 0);
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
          const unsigned short lo = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(hdT) + (int64_t)(cb + min(q, L - 1)) * P + p);
          const unsigned short hi = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(hdT) + (int64_t)(cb + min(q + 1, L - 1)) * P + p);
          hd_early[pl - 1][q >> 1] = (uint32_t)lo | ((uint32_t)hi << 16);
        }
      }
      asm volatile("" ::: "memory");  // (the requests stay up here)
    }
#pragma unroll
    for (int plane = 0; plane < 3; ++plane) {
      const int L = fd.hd[plane].n_levels;
      if (USE_HDT && plane > 0) {  // xz / yz: evaluated by dynhash_fwd_lds_kernel from LDS-resident slice tables
        for (int l0 = 0; l0 < L; l0 += 8) {  // (eight columns requested together, as the static grid's above)
          uint32_t w[8];
          if (ENC_HD_EARLY && l0 == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = (hd_early[plane - 1][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
          } else {
            unsigned short v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
              v[q] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(hdT) + (int64_t)(col - col_dyn0 + min(l0 + q, L - 1)) * P + p);
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = v[q];
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(w[q]));
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (l0 + q < L) row[col + l0 + q] = __builtin_bit_cast(half_t, (unsigned short)w[q]);
        }
        col += L;
        continue;
      }
      const int ca = plane == 2 ? 1 : 0, cb = plane == 0 ? 1 : 2;  // xy, xz, yz
      const float q0[2] = {x0[ca], x0[cb]}, q1[2] = {x1[ca], x1[cb]}, q2[2] = {x2[ca], x2[cb]};
      if (fd.hd_pairs[plane]) {  // pair tables: fetch corners once per distinct cell (field_dev.h)
        const GridDesc& g = fd.hd[plane];
        const PairSel ps0 = pair_sel(tc0.sp, fd.n_slices), ps1 = pair_sel(tc1.sp, fd.n_slices), ps2 = pair_sel(tc2.sp, fd.n_slices);
        const uint4* base = reinterpret_cast<const uint4*>(fd.hd_pairs[plane]);
        const size_t E = fd.hd_entries[plane];
        // The level loop is software-pipelined: the NEXT level's frame-0 corner entries are requested before the current level is
        // evaluated (16 registers), so that a level's gather round trip (L2: ~1.5 k cycles) runs under the previous level's three
        // frame evaluations instead of in front of its own (tools/phase_probe.py enc: this block was 55 % of a group's cycles).
        Cell<2> c0n = locate<2>(q0, g.scale[0]);
        PairCorners pcn;
        pair_fetch(base + (size_t)ps0.q * E + g.offset[0], g, 0, c0n, false, pcn);
        for (int lvl = 0; lvl < L; ++lvl) {
          const Cell<2> c0 = c0n;
          const PairCorners pc = pcn;
          {
            const int ln = min(lvl + 1, L - 1);  // (the last level requests itself once more: an unconditional request, never used)
            c0n = locate<2>(q0, g.scale[ln]);
            pair_fetch(base + (size_t)ps0.q * E + g.offset[ln], g, ln, c0n, false, pcn);
          }
          const float r0 = pair_eval(pc, c0, tc0, ps0.hi);
          float r1 = r0, r2 = r0;
          if (has_fwd) {
            const Cell<2> c1 = locate<2>(q1, g.scale[lvl]);
            const bool same = ps1.q == ps0.q && c1.cell[0] == c0.cell[0] && c1.cell[1] == c0.cell[1];
            PairCorners p1 = pc;
            pair_fetch(base + (size_t)ps1.q * E + g.offset[lvl], g, lvl, c1, same, p1);
            r1 = pair_eval(p1, c1, tc1, ps1.hi);
          }
          if (has_bwd) {
            const Cell<2> c2 = locate<2>(q2, g.scale[lvl]);
            const bool same = ps2.q == ps0.q && c2.cell[0] == c0.cell[0] && c2.cell[1] == c0.cell[1];
            PairCorners p2 = pc;
            pair_fetch(base + (size_t)ps2.q * E + g.offset[lvl], g, lvl, c2, same, p2);
            r2 = pair_eval(p2, c2, tc2, ps2.hi);
          }
          row[col + lvl] = f2h(0.5f * r0 + 0.25f * (r1 + r2));
        }
        col += L;
        ENC_CLK(7)  // (a stack gathered here through its pair tables: xy)
        continue;
      }
      for (int lvl = 0; lvl < L; ++lvl) {
        const float r0 = hash_t_level(fd, plane, lvl, tc0, q0);
        const float r1 = has_fwd ? hash_t_level(fd, plane, lvl, tc1, q1) : r0;
        const float r2 = has_bwd ? hash_t_level(fd, plane, lvl, tc2, q2) : r0;
        row[col + lvl] = f2h(0.5f * r0 + 0.25f * (r1 + r2));
      }
      col += L;
    }
  }
  for (; col < in_pad; ++col) row[col] = (half_t)1.0f;  // tcnn pads the network input with ones
  ENC_CLK(4)  // dynamic hash: xy stack gathered + evaluated, xz / yz columns loaded

  stage_sync();  // rows complete (and visible) before the cooperative copy-out
  if (!SIGMA) {
    copy_out(colsA, in_pad - colsA);
    return;
  }
  copy_out(0, in_pad);
  ENC_CLK(5)  // row copy-out to X
  // ---- density network on the staged rows: mlp_fwd_kernel<8, 1>'s chain, four tiles of 16 rows per wavefront ----
  const int i = lane & 15, g = lane >> 4;
  auto FR = [&](int f) -> h8 { const uint4 u = reinterpret_cast<const uint4*>(stage_all)[f * 64 + lane]; return *reinterpret_cast<const h8*>(&u); };
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    const half_t* xr = wstage + (16 * t + i) * ENC_PITCH + 8 * g;
    h8 xb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const h8*>(xr + 32 * ks);
    f4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      acc[mt] = f4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc[mt] = MFMA(FR(mt * 4 + ks), xb[ks], acc[mt]);
    }
    const h8 hb0 = relu_pack(acc[0], acc[1]), hb1 = relu_pack(acc[2], acc[3]);
    const int64_t orow = wave_p0 + 16 * t + i;
    const bool ok = orow < P;
    if (so.act && ok) {
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, hb0), reinterpret_cast<u32x4_nt*>(so.act + orow * HID + 8 * g));
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4_nt, hb1), reinterpret_cast<u32x4_nt*>(so.act + orow * HID + 32 + 8 * g));
    }
    f4 o = f4{0, 0, 0, 0};
    o = MFMA(FR(16), hb0, o);
    o = MFMA(FR(17), hb1, o);
    if (ok) {
      h4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = f2h(clamp_h(o[r]));
      __builtin_nontemporal_store(__builtin_bit_cast(u32x2_nt, ov), reinterpret_cast<u32x2_nt*>(so.y + orow * 16 + 4 * g));
      if (g == 0) __builtin_nontemporal_store(expf(h2f(ov[0])), so.sigma + orow);
    }
  }
  ENC_CLK(6)  // density network on the staged rows, its stores
  if (!persistent) break;
  stage_sync();  // this group's reads of the staged rows are done before the next group's rows overwrite them
  }  // groups of this wavefront
  ENC_CLK_FLUSH
}

// ---- dynamic hash, forward, with LDS-resident slice tables ---------------------------------------------------
// 576 of the 832 table gathers per sample are the 2-D x time stacks (3 frames x 3 planes x 2 slices x 8 levels x 4
// corners), and through L1/L2 every one of them is its own cache line (one line per clock per CU).  For the xz and yz
// stacks a slice-level is 64 KB, so one workgroup stages BOTH time slices of a (plane, level) in LDS (128 KB) and
// serves the corner reads of its whole chunk of samples x 3 frames from there: 384 of the 576 gathers leave the
// L1/L2 path.  (An xy slice-level is 256 KB: it would need the per-slice interpolated feature -- which tiny-cuda-nn
// rounds to fp16, rounding point R2 -- carried across table parts; xy stays on the direct path in the encode kernel.)
// Frames whose slice pair differs from the current frame's (only when a slice boundary lies between the neighbour
// times) take the direct global path.  Output: column-major hdT[col][P] fp16, merged into X by the encode kernel.
#define DH_THREADS 1024
#define DH_MAX_ENTRIES 8192  // per slice: 2 slices x 8192 x 8 B = 128 KB
// Coordinates for the LDS kernel, as dense arrays: xs[3][P] fp32 (the sample point) and flowT[6][P] fp16 (the two flow vectors,
// exactly the halfs the flow network produced: x1 = x + (float)flow stays bit-identical).  Every (plane, level) task streams its
// samples' coordinates once -- 2 x 4 + 4 x 2 = 16 bytes per sample and task from these arrays, instead of the 48 bytes of whole
// xt / flow rows (or 24 as three fp32 frames per axis): the kernel moves 16 x that and this stream is a third of its time.
__global__ void __launch_bounds__(256) warp_coords_kernel(const float* __restrict__ xt, const half_t* __restrict__ flow16, int64_t P,
                                                         float* __restrict__ xs, half_t* __restrict__ flowT) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float4_t c4 = *reinterpret_cast<const float4_t*>(xt + p * 4);
  const uint4 u = *reinterpret_cast<const uint4*>(flow16 + p * 16);
  const half_t* fh = reinterpret_cast<const half_t*>(&u);
#pragma unroll
  for (int a = 0; a < 3; ++a) xs[(int64_t)a * P + p] = c4[a];
#pragma unroll
  for (int k = 0; k < 6; ++k) flowT[(int64_t)k * P + p] = fh[k];
}

__global__ void __launch_bounds__(DH_THREADS) dynhash_fwd_lds_kernel(FieldDesc fd, const float* __restrict__ xs,
                                                                    const half_t* __restrict__ flowT,
                                                                    const float* __restrict__ tinfo, int64_t P, int64_t chunk,
                                                                    half_t* __restrict__ hdT) {
  extern __shared__ uint4 lds_tab[];  // [entry] = {slice i1: 4 halfs, slice i2: 4 halfs}: both slices of a corner in ONE ds_read_b128
  // task = (plane in {xz, yz}, level); hdT column = levels(xy) + ...
  int task = blockIdx.y, plane = 1;
  if (task >= fd.hd[1].n_levels) { task -= fd.hd[1].n_levels; plane = 2; }
  const int lvl = task;
  const int col = fd.hd[0].n_levels + (plane == 2 ? fd.hd[1].n_levels : 0) + lvl;
  const int ca = plane == 2 ? 1 : 0, cb = 2;
  const GridDesc& g = fd.hd[plane];
  const float scale = g.scale[lvl];
  const uint32_t res = g.res[lvl], size = g.size[lvl];
  const bool hashed = (g.hashed_mask >> lvl) & 1u;
  TimeCoef tc[3] = {time_coef(tinfo[0], fd.n_slices), time_coef(tinfo[1], fd.n_slices), time_coef(tinfo[2], fd.n_slices)};
  const bool has_e[3] = {true, tinfo[3] != 0.0f, tinfo[4] != 0.0f};
  bool in_lds[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) in_lds[e] = has_e[e] && tc[e].sp.i1 == tc[0].sp.i1 && tc[e].sp.i2 == tc[0].sp.i2;
  const bool two = tc[0].sp.i1 != tc[0].sp.i2;
  // stage the current frame's slice pair of this (plane, level): lds_tab[entry] = {slice i1, slice i2}
  if (fd.hd_pairs[plane]) {  // pair-interleaved copy: already in the LDS layout, 16 bytes per lane, coalesced
    const PairSel ps = pair_sel(tc[0].sp, fd.n_slices);
    const uint4* src = reinterpret_cast<const uint4*>(fd.hd_pairs[plane]) + (size_t)ps.q * fd.hd_entries[plane] + g.offset[lvl];
    for (uint32_t i = threadIdx.x; i < size; i += DH_THREADS) {
      uint4 v = src[i];
      if (!two && ps.hi) v = make_uint4(v.z, v.w, v.z, v.w);  // single slice in the pair's high half: serve it as "slice i1"
      lds_tab[i] = v;
    }
  } else {
    const uint2* t1p = reinterpret_cast<const uint2*>(fd.hd_tables[plane][tc[0].sp.i1] + (size_t)g.offset[lvl] * 4);
    const uint2* t2p = reinterpret_cast<const uint2*>(fd.hd_tables[plane][tc[0].sp.i2] + (size_t)g.offset[lvl] * 4);
    for (uint32_t i = threadIdx.x * 2; i < size; i += DH_THREADS * 2) {  // sizes are multiples of 8 entries
      const uint4 a2 = *reinterpret_cast<const uint4*>(&t1p[i]);  // entries i, i+1 of slice i1
      uint4 b2 = a2;
      if (two) b2 = *reinterpret_cast<const uint4*>(&t2p[i]);
      lds_tab[i] = make_uint4(a2.x, a2.y, b2.x, b2.y);
      lds_tab[i + 1] = make_uint4(a2.z, a2.w, b2.z, b2.w);
    }
  }
  __syncthreads();
  half_t* out = hdT + (int64_t)col * P;
  const int64_t lo_p = (int64_t)blockIdx.x * chunk, hi_p = min(P, lo_p + chunk);
  // one sample: the three frames' lookups from the staged table, blended.  FAST (block-uniform: hashed level, power-of-two table --
  // every level of the reference configuration): the index is two multiplies and a mask, without the generic form's per-lookup
  // addressing decision (the kernel is bound by its VALU instructions: SQ_ACTIVE_INST_VALU 30 % of every wave's cycles x 4 waves).
  const bool fast_level = hashed && is_pow2(size);
  auto eval = [&](auto fast_tag, const float xa[3], const float xb[3]) -> half_t {
    constexpr bool FAST = decltype(fast_tag)::value;
    float r[3];
    PairCorners pc0;  // frame 0's corner entries: reused by a warped frame whose point lies in the same cell
    Cell<2> c0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      r[e] = 0.0f;
      if (!has_e[e]) continue;
      const float q[2] = {xa[e], xb[e]};
      if (!in_lds[e]) {  // neighbour frame on another slice pair (rare): direct gathers
        r[e] = hash_t_level(fd, plane, lvl, tc[e], q);
        continue;
      }
      const Cell<2> c = locate<2>(q, scale);
      PairCorners pc = pc0;
      const bool same = e > 0 && c.cell[0] == c0.cell[0] && c.cell[1] == c0.cell[1];
#pragma unroll
      for (int cn = 0; cn < 4; ++cn) {
        uint32_t gv[2];
        (void)corner<2>(c, cn, gv);
        if (!same) pc.e[cn] = lds_tab[FAST ? grid_index_fast<2>(gv, size - 1u) : grid_index<2>(gv, res, size, hashed)];
      }
      if (e == 0) {
        pc0 = pc;
        c0 = c;
      }
      r[e] = pair_eval(pc, c, tc[e], false);  // the staged entry = {slice i1, slice i2}; a single slice sits in the low half
    }
    const float r1 = has_e[1] ? r[1] : r[0], r2 = has_e[2] ? r[2] : r[0];
    return f2h(0.5f * r[0] + 0.25f * (r1 + r2));
  };
  // two samples per iteration: both samples' global loads are in flight before the first LDS lookup
  const float* xa_p = xs + (int64_t)ca * P;
  const float* xb_p = xs + (int64_t)cb * P;
  auto load = [&](int64_t p, float xa[3], float xb[3]) {
    const float a = xa_p[p], b = xb_p[p];
    xa[0] = a;
    xb[0] = b;
#pragma unroll
    for (int e = 1; e < 3; ++e) {  // the warped points: x + flow[:3] (frame 1), x + flow[3:] (frame 2)
      xa[e] = a + h2f(flowT[(int64_t)((e - 1) * 3 + ca) * P + p]);
      xb[e] = b + h2f(flowT[(int64_t)((e - 1) * 3 + cb) * P + p]);
    }
  };
  auto walk = [&](auto fast_tag) {
    for (int64_t p0 = lo_p + threadIdx.x; p0 < hi_p; p0 += 2 * DH_THREADS) {
      const int64_t p1 = p0 + DH_THREADS;
      const bool ok1 = p1 < hi_p;
      const int64_t q1 = ok1 ? p1 : p0;
      float xa0[3], xb0[3], xa1[3], xb1[3];
      load(p0, xa0, xb0);
      load(q1, xa1, xb1);
      out[p0] = eval(fast_tag, xa0, xb0);
      if (ok1) out[p1] = eval(fast_tag, xa1, xb1);
    }
  };
  if (fast_level) walk(std::true_type{});
  else walk(std::false_type{});
}

// ---- sampling that also emits the normalised (x, t) rows the field kernels read --------------------
__global__ void __launch_bounds__(256) sample_rays_xt_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            const float* __restrict__ lin, const float* __restrict__ noise,
                                                            const float* __restrict__ t, int64_t N, int T, float near,
                                                            float far, float bound, float* __restrict__ z_vals,
                                                            float* __restrict__ xt) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * T) return;
  const int64_t ray = idx / T;
  const int ti = (int)(idx - ray * T);
  float z = near + (far - near) * lin[ti];
  if (noise) {
    const float sample_dist = (far - near) / (float)T;
    z = z + (noise[idx] - 0.5f) * sample_dist;
  }
  z_vals[idx] = z;
  float4_t o;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = rays_o[ray * 3 + k] + rays_d[ray * 3 + k] * z;
    v = fminf(fmaxf(v, -bound), bound);          // renderer.py:89
    o[k] = (v + bound) / (2.0f * bound);         // lidar4d.py:141
  }
  o[3] = *t;
  *reinterpret_cast<float4_t*>(xt + idx * 4) = o;
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int l4d_field_width(const l4d_field_desc* f) {
  return 2 * f->n_scales * f->plane_channels + f->hash_static.n_levels * f->hash_static.n_features +
         f->hash_dynamic[0].n_levels + f->hash_dynamic[1].n_levels + f->hash_dynamic[2].n_levels;
}

extern "C" int l4d_time_setup(const float* t, int32_t num_frames, float* tinfo, void* stream) {
  L4D_LAUNCH(time_setup_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, num_frames, tinfo);
  L4D_LAUNCH_CHECK("l4d_time_setup");
  return 0;
}

extern "C" int l4d_sample_rays_xt(const float* rays_o, const float* rays_d, const float* lin, const float* noise,
                                  const float* t, int64_t N, int32_t T, float near, float far, float bound, float* z_vals,
                                  float* xt, void* stream) {
  if (N == 0) return 0;
  L4D_LAUNCH(sample_rays_xt_kernel, dim3((unsigned)ceil_div64(N * T, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, lin, noise, t, N, T, near, far, bound, z_vals, xt);
  L4D_LAUNCH_CHECK("l4d_sample_rays_xt");
  return 0;
}

// pairs[q][e] = {slice q entry e, slice q + 1 entry e}: one thread per (pair, entry)
struct PairSrc {
  const half_t* t[MAX_SLICES];
};
__global__ void __launch_bounds__(256) dyn_pairs_kernel(PairSrc tabs, int n_pairs, int64_t n_entries, uint4* __restrict__ pairs) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  if (e >= n_entries) return;
  const uint2 lo = reinterpret_cast<const uint2*>(tabs.t[q])[e];
  const uint2 hi = reinterpret_cast<const uint2*>(tabs.t[q + 1])[e];
  pairs[(int64_t)q * n_entries + e] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}

extern "C" int l4d_dyn_pairs_build(const void* const* slice_tables, int32_t n_slices, int64_t n_entries, void* pairs, void* stream) {
  if (n_slices < 2 || n_entries == 0) return 0;
  if (n_slices > MAX_SLICES) { l4d_set_error(1, "l4d_dyn_pairs_build: too many slices"); return 1; }
  PairSrc tabs;
  for (int i = 0; i < MAX_SLICES; ++i) tabs.t[i] = i < n_slices ? (const half_t*)slice_tables[i] : nullptr;
  L4D_LAUNCH(dyn_pairs_kernel, dim3((unsigned)ceil_div64(n_entries, 256), n_slices - 1), dim3(256), 0, (hipStream_t)stream, tabs,
             n_slices - 1, n_entries, (uint4*)pairs);
  L4D_LAUNCH_CHECK("l4d_dyn_pairs_build");
  return 0;
}

// hd_scratch of l4d_density_encode_fwd: the LDS kernel's output columns [n_dyn][P] fp16, then the coordinates [3][P] fp32 and flow vectors [6][P] fp16
// ... then (round 5) the static grid's level-major columns [n_levels][P][4] fp16
static inline int64_t enc_ws_hs_offset(const l4d_field_desc* f, int64_t P) {
  const int64_t n_dyn = f->hash_dynamic[0].n_levels + f->hash_dynamic[1].n_levels + f->hash_dynamic[2].n_levels;
  return (n_dyn * P * 2 + 255) / 256 * 256 + (3 * P * 4 + 255) / 256 * 256 + (6 * P * 2 + 255) / 256 * 256;
}
// (the static grid's columns only where the pre-pass can run: F = 4 and enough points -- 0.8 GB at 12.6 M samples otherwise unused)
extern "C" int64_t l4d_density_encode_fwd_workspace(const l4d_field_desc* f, int64_t P) {
  const bool hs_cols = f->hash_static.n_features == 4 && P >= ENC_SPLIT_MIN_POINTS;
  return enc_ws_hs_offset(f, P) + (hs_cols ? (int64_t)f->hash_static.n_levels * P * 8 : 0);
}

L4D_INTERNAL int l4d_hs_pairld();
L4D_INTERNAL int l4d_hs_pair_ok(const GridDesc* g, int n_features, const void* table);
L4D_INTERNAL int l4d_hashgrid_levels_launch(const GridDesc* g, int n_dims, int n_features, const float* x, int64_t P, int x_stride,
                                            const int* cols3, const void* table, void* lvlT, void* stream);
// static grid through the level-major pre-pass (default; L4D_ENC_HS_SPLIT=0: gathered inside the encode kernel as in rounds 1-4)
static int enc_hs_split() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("L4D_ENC_HS_SPLIT"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}

extern "C" int64_t l4d_plane_rows_workspace(const l4d_field_desc* f) {
  int64_t n = 0;
  for (int s = 0; s < f->n_scales; ++s)
    for (int j = 0; j < 3; ++j) n += (int64_t)TROWS_FRAMES * f->plane_res[s * 4 + j] * f->plane_channels;
  return n * (int64_t)sizeof(float);
}

extern "C" int l4d_mlp_fwd_sigma(const void* x, int64_t P, int32_t in_pad, int32_t n_hidden, const void* weights, void* y, void* act,
                                 float* sigma, void* stream);
// density network inside the encode kernel (default; L4D_ENC_SIGMA=0: a launch of its own behind it, A/B)
static int enc_sigma_fused() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("L4D_ENC_SIGMA"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}
// so: null, or where the density network's forward pass (n_hidden hidden layers) puts its outputs
static int encode_fwd_impl(const l4d_field_desc* f, const float* xt, const void* flow16, const float* tinfo,
                           int64_t P, void* X, int32_t in_pad, void* hd_scratch, float* plane_rows, void* stream,
                           const SigmaOut* so, int n_hidden) {
  if (P == 0) return 0;
  FieldDesc d;
  if (make_field(f, d)) return 1;
  if (l4d_field_width(f) > in_pad || in_pad % 8 || in_pad > ENC_MAX_IN_PAD) {
    l4d_set_error(1, "l4d_density_encode_fwd: in_pad too small for the field width (or not a multiple of 8)");
    return 1;
  }
  hipStream_t main_s = (hipStream_t)stream;
  // side stream: the LDS evaluation of the xz / yz stacks runs next to the plane part of the encode (l4d_streams_config bit 0)
  // static grid: level-major pre-pass into the workspace (needs the workspace, F = 4 and the one-kernel form of the encode);
  // l4d_streams_config bit 2: that pre-pass (L2 gathers) runs NEXT TO the LDS kernel (VALU / LDS-bound) instead of behind it
  const bool hs_pre = hd_scratch && plane_rows && !(l4d_streams_mask() & 1) && enc_hs_split() && f->hash_static.n_features == 4 && P >= ENC_SPLIT_MIN_POINTS;
  const bool hs_side = hs_pre && (l4d_streams_mask() & 4);
  const bool split = hd_scratch && (l4d_streams_mask() & 1) && P >= ENC_SPLIT_MIN_POINTS;
  hipStream_t dh_s = main_s;
  if (split || hs_side) {
    dh_s = (hipStream_t)l4d_side_fork(stream, 0);
    if (!dh_s) return 1;
  }
  if (hd_scratch) {
    if (d.hd[1].size[d.hd[1].n_levels - 1] > DH_MAX_ENTRIES || d.hd[2].size[d.hd[2].n_levels - 1] > DH_MAX_ENTRIES) {
      l4d_set_error(1, "l4d_density_encode_fwd: xz/yz slice tables exceed the LDS staging size; pass hd_scratch = null");
      return 1;
    }
    int n_chunks = (int)std::min<int64_t>(256, std::max<int64_t>(1, ceil_div64(P, 8192)));
    const int64_t chunk = ceil_div64(P, n_chunks);
    n_chunks = (int)ceil_div64(P, chunk);
    (void)hipFuncSetAttribute((const void*)dynhash_fwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DH_MAX_ENTRIES * 8);
    const int64_t n_dyn = d.hd[0].n_levels + d.hd[1].n_levels + d.hd[2].n_levels;
    float* xs = (float*)((char*)hd_scratch + (n_dyn * P * 2 + 255) / 256 * 256);
    half_t* flowT = (half_t*)((char*)xs + (3 * P * 4 + 255) / 256 * 256);
    L4D_LAUNCH(warp_coords_kernel, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, dh_s, xt, (const half_t*)flow16, P, xs, flowT);
    L4D_LAUNCH(dynhash_fwd_lds_kernel, dim3(n_chunks, d.hd[1].n_levels + d.hd[2].n_levels), dim3(DH_THREADS),
               2 * DH_MAX_ENTRIES * 8, dh_s, d, xs, flowT, tinfo, P, chunk, (half_t*)hd_scratch);
  }
  half_t* hsT = hs_pre ? (half_t*)((char*)hd_scratch + enc_ws_hs_offset(f, P)) : nullptr;
  if (hs_pre) {
    const int cols3[3] = {0, 1, 2};
    const int rc = l4d_hashgrid_levels_launch(&d.hs, 3, 4, xt, P, 4, cols3, d.hs_table, hsT, main_s);
    if (hs_side && l4d_side_join(stream, 0)) return 1;  // (also on the error path: the launch stream is ordered behind the side stream again)
    if (rc) return 1;
  }
  const PlaneRows pr = make_plane_rows(d, plane_rows);
  if (plane_rows)  // tinfo[0..2] = t, t1, t2: frames without a neighbour get a row nobody reads
    L4D_LAUNCH(plane_time_rows_kernel, dim3(2, d.planes.n_scales * 3, TROWS_FRAMES), dim3(256), 0, main_s, d, pr, tinfo, plane_rows);
  const dim3 egrid((unsigned)xcd_grid(ceil_div64(P, ENC_THREADS)));
  const int colsA = 2 * d.planes.n_scales * 8;
  const int enc_lds = ENC_THREADS * (std::max(colsA, in_pad - colsA) + 8) * 2;
#define ENC_LAUNCH(HDT, ROWS, PART)                                                                                         \
  L4D_LAUNCH((density_encode_fwd_kernel<HDT, ROWS, PART>), egrid, dim3(ENC_THREADS), enc_lds, main_s, d, xt,                \
             (const half_t*)flow16, tinfo, P, (const half_t*)hd_scratch, (half_t*)X, in_pad, pr, (const half_t*)nullptr, no_sigma)
  const SigmaOut no_sigma{nullptr, nullptr, nullptr, nullptr, 0};
  // the density network as the kernel's epilogue: the default network shape behind the level-major form of the encode
  bool sigma_fused = so && hs_pre && in_pad == 128 && n_hidden == 1 && enc_sigma_fused();
  const int lds = ENC_SIGMA_FRAGS * 1024 + ENC_SIGMA_THREADS * (in_pad + 8) * 2;
  // (154 KB of dynamic LDS: a device that does not grant it takes the two-launch path below instead of failing at the launch)
  if (sigma_fused && hipFuncSetAttribute((const void*)density_encode_fwd_kernel<true, true, 0, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
    (void)hipGetLastError();
    sigma_fused = false;
  }
  if (sigma_fused) {
    static int persistent = -1, n_cu = 0;
    if (persistent < 0) {
      const char* e = getenv("L4D_ENC_PERSISTENT");
      persistent = (e && e[0] == '0') ? 0 : 1;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 8) n_cu = 256;
    }
    SigmaOut so_l = *so;
    so_l.persistent = persistent;
    const unsigned grid = persistent ? (unsigned)(n_cu / 8 * 8) : (unsigned)xcd_grid(ceil_div64(P, ENC_SIGMA_THREADS));  // one workgroup per CU (LDS)
    L4D_LAUNCH((density_encode_fwd_kernel<true, true, 0, 1, true>), dim3(grid), dim3(ENC_SIGMA_THREADS), lds,
               main_s, d, xt, (const half_t*)flow16, tinfo, P, (const half_t*)hd_scratch, (half_t*)X, in_pad, pr, (const half_t*)hsT, so_l);
  } else if (hs_pre) {
    L4D_LAUNCH((density_encode_fwd_kernel<true, true, 0, 1>), egrid, dim3(ENC_THREADS), enc_lds, main_s, d, xt,
               (const half_t*)flow16, tinfo, P, (const half_t*)hd_scratch, (half_t*)X, in_pad, pr, (const half_t*)hsT, no_sigma);
  } else if (!split && hd_scratch && plane_rows && l4d_hs_pair_ok(&d.hs, f->hash_static.n_features, d.hs_table)) {
    L4D_LAUNCH((density_encode_fwd_kernel<true, true, 0, 2>), egrid, dim3(ENC_THREADS), enc_lds, main_s, d, xt,
               (const half_t*)flow16, tinfo, P, (const half_t*)hd_scratch, (half_t*)X, in_pad, pr, (const half_t*)nullptr, no_sigma);
  } else if (split) {  // plane columns while the side stream evaluates the xz / yz stacks, then the hash columns
    if (plane_rows) ENC_LAUNCH(true, true, 1);
    else ENC_LAUNCH(true, false, 1);
    if (l4d_side_join(stream, 0)) return 1;
    if (plane_rows) ENC_LAUNCH(true, true, 2);
    else ENC_LAUNCH(true, false, 2);
  } else if (hd_scratch && plane_rows) ENC_LAUNCH(true, true, 0);
  else if (hd_scratch) ENC_LAUNCH(true, false, 0);
  else if (plane_rows) ENC_LAUNCH(false, true, 0);
  else ENC_LAUNCH(false, false, 0);
#undef ENC_LAUNCH
  L4D_LAUNCH_CHECK("l4d_density_encode_fwd");
  if (so && !sigma_fused)  // other shapes / paths: the network as a launch of its own on the rows just written
    return l4d_mlp_fwd_sigma(X, P, in_pad, n_hidden, so->w, so->y, so->act, so->sigma, stream);
  return 0;
}

extern "C" int l4d_density_encode_fwd(const l4d_field_desc* f, const float* xt, const void* flow16, const float* tinfo,
                                      int64_t P, void* X, int32_t in_pad, void* hd_scratch, float* plane_rows, void* stream) {
  return encode_fwd_impl(f, xt, flow16, tinfo, P, X, in_pad, hd_scratch, plane_rows, stream, nullptr, 0);
}

// l4d_density_encode_fwd + l4d_mlp_fwd_sigma (the density network, lidar4d.py:181-186) in one call: for the default network shape
// (128 -> 64 -> 16) behind the level-major encode the network runs as the encode kernel's epilogue on the rows in LDS
extern "C" int l4d_density_encode_sigma_fwd(const l4d_field_desc* f, const float* xt, const void* flow16, const float* tinfo,
                                            int64_t P, void* X, int32_t in_pad, void* hd_scratch, float* plane_rows,
                                            const void* sigma_weights, int32_t n_hidden, void* y, void* act, float* sigma, void* stream) {
  if (!sigma_weights || !y || !sigma) { l4d_set_error(1, "l4d_density_encode_sigma_fwd: weights / y / sigma is null"); return 1; }
  const SigmaOut so{(const half_t*)sigma_weights, (half_t*)y, (half_t*)act, sigma, 0};
  return encode_fwd_impl(f, xt, flow16, tinfo, P, X, in_pad, hd_scratch, plane_rows, stream, &so, n_hidden);
}
