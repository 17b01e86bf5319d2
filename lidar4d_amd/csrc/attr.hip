// The two attribute networks of LiDAR4D.attribute (model/lidar4d.py:191-223: raydrop_net and intensity_net, tcnn FullyFusedMLP
// 96 -> 64 -> 64 -> 1, evaluated on ONE input h = [frequency encoding of the ray direction (72) | geo_feat (15) | padding]) on the
// compacted work list of samples with weight > 1e-4 -- with the direction encoding HOISTED out of the per-sample product.
//
// The 72 encoding columns are the same for every sample of a ray (up to 768 of them), so their share of the first layer,
//     e[ray] = W1[:, 0:72] . enc(d_ray)                                   (one 64-vector per ray and network, fp32),
// is computed once per ray (attr_ray_term_kernel: 16,384 x 72 x 64 multiply-adds per network instead of 11.9 M x 72 x 64) and
// enters the matrix-core chain of mlp.hip as the INITIAL VALUE of the first layer's accumulator; what is left per sample is a
// K = 32 step over [1, g0 .. g14 | ones x 8 | 0 x 8] (the sigma network's output row, as mlp.hip's gathered kernels read it).  In
// the backward pass the encoding's weight gradient is (sum over the ray's rows of dZ1)^T x enc(d_ray): the row sums are formed on
// the matrix cores as well (dZ1^T times an indicator column per ray), carried in 16 accumulator registers across the consecutive
// tiles of a ray and flushed once per ray; attr_enc_grad_kernel turns the ~20 k flushed sums into dW1[:, 0:72].  First-layer K
// 96 -> 32, dW1 accumulators 96 -> 32 registers: the backward kernel fits 256 registers, i.e. TWO wavefronts per SIMD -- a lone
// wavefront issues one VALU instruction per ~5 cycles on this chip, two share the SIMD at ~2.5-3.5 (profiles/r06_ubench_valu.txt),
// and the register-resident backward kernels are bound by exactly that.
//
// Summation order: the fp32 accumulation of a first-layer pre-activation is e (72 products, ascending column) + one MFMA step
// instead of three MFMA steps -- equal to the last bit of fp32 or not, like any two MFMA tilings; the fp16 rounding points are
// mlp.hip's.
//
// Work-list contract (l4d_composite_fwd_padded, render.hip): every maximal run of entries of one ray is at least 32 long (padded
// with -1 = "no sample"), so that a 32-row tile meets at most two rays.
#include "common.h"
#include "wave_dev.h"
#include "mlp_dev.h"

#define ATTR_IN 96     // logical input width of the networks (weight row stride)
#define ATTR_ENC 72    // direction-encoding columns (Frequency, degree 12, 3 inputs: SURVEY A.2)
#define ATTR_NF_FWD 14  // forward fragments per network: 4 (layer 1, K = 32) + 8 (layer 2) + 2 (output)

struct AttrNets {
  const int32_t* idx;    // work list, -1 = padding
  const int32_t* count;  // rows of the work list (padding included)
  const half_t* h;       // [samples, 16] sigma-network output rows
  const float* e;        // [2][rays][64] per-ray first-layer terms (network 0 = raydrop, 1 = intensity)
  int64_t cap;
  int32_t n_rays, T;
};

// ---- per-ray first-layer term -------------------------------------------------------------------
// e[net][ray][n] = sum_{c < 72} W1_net[n][c] * enc[ray][c]: fp16 operands, fp32 accumulation in ascending column order.
__global__ void __launch_bounds__(256) attr_ray_term_kernel(const half_t* __restrict__ w_r, const half_t* __restrict__ w_i,
                                                          const half_t* __restrict__ enc, int32_t n_rays, float* __restrict__ e) {
  __shared__ half_t w_s[2][HID][ATTR_ENC + 2];  // (+2: consecutive neurons on different banks)
  for (int q = threadIdx.x; q < 2 * HID * ATTR_ENC; q += blockDim.x) {
    const int net = q / (HID * ATTR_ENC), n = (q / ATTR_ENC) % HID, c = q % ATTR_ENC;
    w_s[net][n][c] = (net ? w_i : w_r)[n * ATTR_IN + c];
  }
  __syncthreads();
  const int n = threadIdx.x & 63;
  for (int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)gridDim.x * 4) {
    const half_t* er = enc + ray * ATTR_ENC;  // (the 64 lanes of a wavefront read the same 144 bytes)
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 8
    for (int c = 0; c < ATTR_ENC; ++c) {
      const float x = h2f(er[c]);
      a0 = fmaf(h2f(w_s[0][n][c]), x, a0);
      a1 = fmaf(h2f(w_s[1][n][c]), x, a1);
    }
    e[ray * HID + n] = a0;
    e[((int64_t)n_rays + ray) * HID + n] = a1;
  }
}

// B fragment of the per-sample K = 32 step for lane (i, g): columns 8g .. 8g + 7 of [1, g0 .. g14 | ones x 8 | 0 x 8]
__device__ __forceinline__ const uint4* attr_x_ptr(const half_t* h, int64_t p, int g) { return reinterpret_cast<const uint4*>(h + p * 16 + 8 * (g & 1)); }
__device__ __forceinline__ uint4 attr_x_fix(int g, uint4 u) {
  if (g >= 2) return make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);  // ones (the weights of columns 96 .. 103 are zero)
  if (g == 0) u.x = (u.x & 0xFFFF0000u) | 0x3C00u;  // the sigma logit's slot carries the constant
  return u;
}
// accumulator start of the chain tile for the lane's row: neurons perm_row(mt, 4g + r) = 32 (mt >> 1) + 8g + 4 (mt & 1) + r
__device__ __forceinline__ void attr_e_load(const float* e_ray, int g, f4 acc[4]) {
  acc[0] = *reinterpret_cast<const f4*>(e_ray + 8 * g);
  acc[1] = *reinterpret_cast<const f4*>(e_ray + 8 * g + 4);
  acc[2] = *reinterpret_cast<const f4*>(e_ray + 32 + 8 * g);
  acc[3] = *reinterpret_cast<const f4*>(e_ray + 32 + 8 * g + 4);
}
// forward fragments of one network into LDS (layout of mlp_fwd_kernel with K = 32 in layer 1: the weight columns 72 .. 103 in the
// physical order of col_map(., 72); columns >= 96 do not exist and read as zero)
__device__ __forceinline__ void attr_build_fwd_frags(const half_t* __restrict__ w, uint4 (*frags)[64], int wave, int n_waves, int lane) {
  const int i = lane & 15, g = lane >> 4;
  for (int f = wave; f < ATTR_NF_FWD; f += n_waves) {
    h8 v;
    if (f < 4) v = build_frag(w, HID, ATTR_IN, 0, perm_row(f, i), ATTR_ENC + 8 * g, ATTR_ENC);
    else if (f < 12) v = build_frag(w + HID * ATTR_IN, HID, HID, 0, perm_row((f - 4) / 2, i), 32 * ((f - 4) % 2) + 8 * g);
    else v = build_frag(w + HID * ATTR_IN + HID * HID, 16, HID, 0, i, 32 * (f - 12) + 8 * g);
    frags[f][lane] = *reinterpret_cast<uint4*>(&v);
  }
}

// ---- forward: both networks on a 16-row tile ------------------------------------------------------
// attr_dense[sample] = (sigmoid(raydrop), sigmoid(intensity)) rounded to fp16 like the reference's (lidar4d.py:210-219), and the
// same pair at attr_compact[row] for the backward.
__global__ void __launch_bounds__(256) attr_nets_fwd_kernel(AttrNets a, const half_t* __restrict__ w_r, const half_t* __restrict__ w_i,
                                                          float* __restrict__ attr_dense, float* __restrict__ attr_compact) {
  __shared__ uint4 frags[2][ATTR_NF_FWD][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  attr_build_fwd_frags(w_r, frags[0], wave, 4, lane);
  attr_build_fwd_frags(w_i, frags[1], wave, 4, lane);
  __syncthreads();
  const int64_t P = min((int64_t)*a.count, a.cap);
  const int64_t n_tiles = (P + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * 4;
  auto FR = [&](int net, int f) -> h8 { uint4 u = frags[net][f][lane]; return *reinterpret_cast<h8*>(&u); };
  // the work-list entry of a tile is read one tile ahead: the row and e loads depend on it
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  auto entry_of = [&](int64_t t) -> int32_t {
    const int64_t row = min(t, n_tiles - 1) * 16 + i;
    return a.idx[row < P ? row : 0];
  };
  int32_t ent_nxt = tile < n_tiles ? entry_of(tile) : -1;
  for (; tile < n_tiles; tile += stride) {
    const int32_t ent = ent_nxt;
    ent_nxt = entry_of(tile + stride);
    const int64_t row = tile * 16 + i;
    const bool ok = row < P && ent >= 0;
    const int64_t p = ok ? (int64_t)ent : 0;
    const int64_t ray = (int64_t)((uint32_t)p / (uint32_t)a.T);
    uint4 xr = *attr_x_ptr(a.h, p, g);
    f4 acc[2][4];
    attr_e_load(a.e + ray * HID, g, acc[0]);
    attr_e_load(a.e + ((int64_t)a.n_rays + ray) * HID, g, acc[1]);
    xr = attr_x_fix(g, xr);
    const h8 xb = *reinterpret_cast<h8*>(&xr);
    float sg[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[net][mt] = MFMA(FR(net, mt), xb, acc[net][mt]);
      h8 hb[2];
      hb[0] = relu_pack(acc[net][0], acc[net][1]);
      hb[1] = relu_pack(acc[net][2], acc[net][3]);
      f4 c[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        c[mt] = MFMA(FR(net, 4 + mt * 2 + 0), hb[0], (f4{0, 0, 0, 0}));
        c[mt] = MFMA(FR(net, 4 + mt * 2 + 1), hb[1], c[mt]);
      }
      hb[0] = relu_pack(c[0], c[1]);
      hb[1] = relu_pack(c[2], c[3]);
      f4 o = MFMA(FR(net, 12), hb[0], (f4{0, 0, 0, 0}));
      o = MFMA(FR(net, 13), hb[1], o);
      const float y0 = h2f(f2h(clamp_h(o[0])));  // output row 0 = lanes g == 0, r == 0
      sg[net] = h2f(f2h(1.0f / (1.0f + expf(-y0))));
    }
    if (ok && g == 0) {
      *reinterpret_cast<float2*>(attr_dense + p * 2) = make_float2(sg[0], sg[1]);
      *reinterpret_cast<float2*>(attr_compact + row * 2) = make_float2(sg[0], sg[1]);
    }
  }
}

// ---- backward: ONE network per launch -----------------------------------------------------------
// Fragments (as mlp_bwd_kernel<.., RECOMP>): W_o^T chain 4 + natural 4; W_2^T chain 8 + natural 8; W_1^T natural for the 16
// geometry columns 2; forward chain 12 (layer 1 K = 32: 4, layer 2: 8).
#define ATTR_BF_WOT_P 0
#define ATTR_BF_WOT_N 4
#define ATTR_BF_W2T 8
#define ATTR_BF_W1T 24
#define ATTR_BF_FWD 26
#define ATTR_NF_BWD 38
#define ATTR_BF_ID 38  // + 2: the identity fragments of the transposes (8 registers if held)
struct AttrBwd {
  const float* d_attr;   // [samples, 2]
  const float* attr_c;   // [rows, 2]
  half_t* dh;            // [samples, 16]: column 0 holds the density activation's adjoint (l4d_sigma_bwd_rows)
  float* grad_w;         // this network's fp32 gradient [64 x 96 | 64 x 64 | 16 x 64]
  float* part;           // flushed row sums: [slots][64 + 16] floats (64 sums, then the ray as an int32, padding)
  int32_t* part_count;   // slots used (this launch appends)
  int32_t part_cap;
  int32_t ch, add;       // channel 0 / 1; add: add to dh's geometry columns (the other network stored them) instead of storing
  float loss_scale, inv_scale;
};
#define ATTR_PART_STRIDE 80
#ifndef ATTR_BWD_WAVES
#define ATTR_BWD_WAVES 1  // wavefronts per SIMD the backward kernel is compiled for (1: rows prefetched a tile ahead; 2: no registers for that)
#endif

__global__ void __launch_bounds__(256, ATTR_BWD_WAVES) attr_net_bwd_kernel(AttrNets a, const half_t* __restrict__ w, AttrBwd b) {
  __shared__ uint4 frags[ATTR_NF_BWD + 2][64];
  // (the wavefront index as a SCALAR: everything derived from it -- the tile range, the loop counter, 64-bit -- otherwise lives in
  // vector registers this kernel does not have)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 15, g = lane >> 4;
  const half_t* W1 = w;
  const half_t* W2 = w + HID * ATTR_IN;
  const half_t* Wo = W2 + HID * HID;
  for (int f = wave; f < ATTR_NF_BWD; f += 4) {
    h8 v;
    if (f < ATTR_BF_W2T) {
      const int mt = f & 3;
      v = build_frag(Wo, 16, HID, 1, f < 4 ? perm_row(mt, i) : 16 * mt + i, 8 * g);
    } else if (f < ATTR_BF_W1T) {
      const int r = f - ATTR_BF_W2T, nat = r / 8, mt = (r % 8) / 2, ks = r % 2;
      v = build_frag(W2, HID, HID, 1, nat ? 16 * mt + i : perm_row(mt, i), 32 * ks + 8 * g);
    } else if (f < ATTR_BF_FWD) {
      v = build_frag(W1, HID, ATTR_IN, 1, ATTR_ENC + i, 32 * (f - ATTR_BF_W1T) + 8 * g, ATTR_ENC);  // rows = physical columns 72 .. 87
    } else if (f < ATTR_BF_FWD + 4) {
      v = build_frag(W1, HID, ATTR_IN, 0, perm_row(f - ATTR_BF_FWD, i), ATTR_ENC + 8 * g, ATTR_ENC);
    } else {
      const int q = f - ATTR_BF_FWD - 4;
      v = build_frag(W2, HID, HID, 0, perm_row(q / 2, i), 32 * (q % 2) + 8 * g);
    }
    frags[f][lane] = *reinterpret_cast<uint4*>(&v);
  }
  if (wave < 2) {
    const h8 v = ident_frag(lane, wave);
    frags[ATTR_BF_ID + wave][lane] = *reinterpret_cast<const uint4*>(&v);
  }
  __syncthreads();
  auto FR = [&](int f) -> h8 { uint4 u = frags[f][lane]; return *reinterpret_cast<h8*>(&u); };
#define I0 FR(ATTR_BF_ID)
#define I1 FR(ATTR_BF_ID + 1)

  // (dy has ONE non-zero column -- the network's single output --, so of dWo = dy^T H2 only row 0 exists: C[m = 0][n = i] of the
  // product, component 0 of the lanes g == 0.  One register per column tile instead of the four of a C fragment.)
  float dWo0[4];
  f4 dW2[4][4], dW1[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dWo0[q] = 0.0f;
    dW1[q][0] = dW1[q][1] = f4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) dW2[q][r] = f4{0, 0, 0, 0};
  }
  const int64_t P = min((int64_t)*a.count, a.cap);
  const int64_t n_macro = (P + 31) / 32;
  // CONTIGUOUS tile ranges per wavefront: the consecutive tiles of a ray then meet in one wavefront, whose cs registers carry the
  // ray's row sums from tile to tile
  const int64_t n_waves = (int64_t)gridDim.x * 4, wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t per = (n_macro + n_waves - 1) / n_waves;
  const int64_t t_lo = min(wid * per, n_macro), t_hi = min(t_lo + per, n_macro);
  // ROW SUMS IN THE PADDING COLUMNS OF dW1.  The second column tile of the per-sample block covers the physical columns 88 .. 103;
  // 96 .. 103 do not exist (their weights read as zero, whatever x holds there).  With x[row][96] = "the row belongs to ray u" and
  // x[row][97] = "... to ray v" the dW1 accumulation itself leaves sum_rows dZ1[row][n] per ray in columns 8 and 9 of that tile --
  // no extra accumulators, no extra MFMA.  u = the ray the wavefront is summing (its consecutive tiles mostly hold one ray), v = a
  // second ray in the same tile (at most two: work-list contract).
  int32_t acc_ray = -1;  // wave-uniform: the ray whose row sums column 8 holds (-1: none)
  auto flush = [&](int col, int32_t ray) {  // column `col` (8 / 9) of dW1[.][1] = lanes i == col: 4 lanes x 16 sums; zeroed afterwards
    int32_t slot = 0;
    if (lane == 0) slot = atomicAdd(b.part_count, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (i == col) {
      if (slot < b.part_cap) {
        float* d = b.part + (int64_t)slot * ATTR_PART_STRIDE;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f4*>(d + 16 * mt + 4 * g) = dW1[mt][1];
        if (g == 0) *reinterpret_cast<int32_t*>(d + 64) = ray;
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) dW1[mt][1] = f4{0, 0, 0, 0};
    }
  };
  // INPUTS OF A TILE, FETCHED AHEAD.  Nothing but the wavefront's own program order hides a memory round trip here (one or two
  // wavefronts per SIMD), and a tile's loads are a chain: work-list entry -> the sample's rows.  The entries are read two tiles
  // ahead; with ATTR_BWD_WAVES == 1 (512 registers) the rows themselves -- x, the two factors of dy, this lane's piece of dh -- one
  // tile ahead (as mlp.hip's backward kernels do); the two-wavefront build has no registers for that and issues them at the head of
  // the tile (measured, session s11 of round 6: both forms 1.55 ms per launch without the row prefetch -- 70 % of the wave cycles
  // waiting for memory).
  constexpr bool PF = ATTR_BWD_WAVES == 1;
  struct TileIn {
    uint4 x[2];
    float d0[2], d1[2];
    uint2 dh[2];
  };
  auto entries_of = [&](int64_t mt, int32_t e[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = (uint32_t)(min(mt, n_macro - 1) * 32) + 8 * (i >> 2) + 4 * q + (i & 3);
      e[q] = a.idx[r < P ? r : 0];
    }
  };
  auto load_tile = [&](int64_t mt, const int32_t e[2], TileIn& t) {  // ISSUES the loads; nothing here touches their destinations
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t r = (uint32_t)(min(mt, n_macro - 1) * 32) + 8 * (i >> 2) + 4 * q + (i & 3);
      const uint32_t rc = r < P ? r : 0u;
#ifdef ATTR_EXP_FAKE_LOADS
      const uint32_t p = e[q] >= 0 ? (uint32_t)(e[q] & 1023) : 0u;  // (timing experiment: every row load hits L1 / L2)
#else
      const uint32_t p = e[q] >= 0 ? (uint32_t)e[q] : 0u;
#endif
      t.x[q] = *attr_x_ptr(a.h, p, g);
      t.d0[q] = b.d_attr[(int64_t)p * 2 + b.ch];
      t.d1[q] = b.attr_c[(int64_t)rc * 2 + b.ch];
      t.dh[q] = *reinterpret_cast<const uint2*>(b.dh + (int64_t)p * 16 + 4 * g);  // (this lane's 4-column piece; g == 0 also keeps column 0)
    }
  };
  int32_t ent_cur[2] = {-1, -1}, ent_nxt[2] = {-1, -1};
  TileIn cur;
  if (t_lo < t_hi) {
    entries_of(t_lo, ent_cur);
    entries_of(t_lo + 1, ent_nxt);
    if (PF) load_tile(t_lo, ent_cur, cur);
  }
  for (int64_t mtile = t_lo; mtile < t_hi; ++mtile) {
    asm volatile("" ::: "memory");  // (the weight fragments are re-read from LDS in every tile: mlp.hip)
    TileIn nxt;
    int32_t ent_nn[2];
    if (PF) load_tile(mtile + 1, ent_nxt, nxt);  // (past the range: the last tile once more, never used)
    else load_tile(mtile, ent_cur, cur);
    entries_of(mtile + 2, ent_nn);
    asm volatile("" ::: "memory");  // keeps the prefetch up here
    // ---- this tile's rows: chain tiles a = 0, 1 with row(a, i) = 8 (i >> 2) + 4a + (i & 3) ----
    uint32_t rows[2], ps[2];  // (cap < 2^31: checked at launch)
    bool ok[2];
    int32_t ray[2];
    uint4 xr[2];
    float dyf[2][2];
    uint2 dh_old[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      rows[q] = (uint32_t)(mtile * 32) + 8 * (i >> 2) + 4 * q + (i & 3);
      const int32_t ent = ent_cur[q];
      ok[q] = rows[q] < P && ent >= 0;
      ps[q] = ok[q] ? (uint32_t)ent : 0u;
      ray[q] = ok[q] ? (int32_t)(ps[q] / (uint32_t)a.T) : -1;
    }
    // Which rays does the tile hold?  (wave-uniform: scalar registers from here on)
    const unsigned long long m0 = __ballot(ok[0]), m1 = __ballot(ok[1]);
    unsigned long long v0 = 0ull, v1 = 0ull;  // rows of the second ray
    int32_t ray_u = -1, ray_v = -1;
    if ((m0 | m1) != 0ull) {
      ray_u = __builtin_amdgcn_readlane(m0 ? ray[0] : ray[1], m0 ? __builtin_ctzll(m0) : __builtin_ctzll(m1));
      v0 = __ballot(ok[0] && ray[0] != ray_u);
      v1 = __ballot(ok[1] && ray[1] != ray_u);
      if ((v0 | v1) != 0ull) ray_v = __builtin_amdgcn_readlane(v0 ? ray[0] : ray[1], v0 ? __builtin_ctzll(v0) : __builtin_ctzll(v1));
      if (acc_ray >= 0 && acc_ray != ray_u) flush(8, acc_ray);
      acc_ray = ray_u;
    }
    const unsigned long long u0 = m0 & ~v0, u1 = m1 & ~v1;  // rows of the first
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xr[q] = cur.x[q];
      dyf[q][0] = cur.d0[q];
      dyf[q][1] = cur.d1[q];
      dh_old[q] = cur.dh[q];
    }
    // ---- forward chain again: hidden activations of both layers ----
    h8 xf[2], h1[2][2], h2[2][2];
    h8 dzf[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xr[q] = attr_x_fix(g, xr[q]);
      if (!ok[q]) xr[q] = make_uint4(0, 0, 0, 0);
      xf[q] = *reinterpret_cast<h8*>(&xr[q]);
      f4 acc[4];
      // (the ray's first-layer term is fetched here, one chain tile at a time: both at the head of the tile were 32 registers that
      // the kernel does not have at two wavefronts per SIMD; the partner wavefront covers the L2 round trip)
      attr_e_load(a.e + ((int64_t)b.ch * a.n_rays + (ray[q] < 0 ? 0 : ray[q])) * HID, g, acc);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA(FR(ATTR_BF_FWD + mt), xf[q], acc[mt]);
      h1[q][0] = relu_pack(acc[0], acc[1]);
      h1[q][1] = relu_pack(acc[2], acc[3]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        acc[mt] = MFMA(FR(ATTR_BF_FWD + 4 + mt * 2 + 0), h1[q][0], (f4{0, 0, 0, 0}));
        acc[mt] = MFMA(FR(ATTR_BF_FWD + 4 + mt * 2 + 1), h1[q][1], acc[mt]);
      }
      h2[q][0] = relu_pack(acc[0], acc[1]);
      h2[q][1] = relu_pack(acc[2], acc[3]);
        // dy[row][0] = d_attr * s (1 - s) * loss_scale (adjoint of the sigmoid + scatter, lidar4d.py:210-219); other columns 0
      const float sg = dyf[q][1];
      const half_t v = f2h_grad(dyf[q][0] * sg * (1.0f - sg) * b.loss_scale);
      const uint32_t dy0 = (g == 0 && ok[q]) ? (uint32_t)__builtin_bit_cast(unsigned short, v) : 0u;
      const uint4 dyu = make_uint4(dy0, 0u, 0u, 0u);
      dzf[q][0] = *reinterpret_cast<const h8*>(&dyu);
    }
    __builtin_amdgcn_sched_barrier(0);  // (nothing of the next step -- its LDS fragment reads above all -- is scheduled up into this one: registers)
    // ---- output layer ----
    h8 dzT[4], hT[4], dyT;
    {
      f4 t0 = MFMA(dzf[0][0], I0, (f4{0, 0, 0, 0}));
      f4 t1 = MFMA(dzf[1][0], I0, (f4{0, 0, 0, 0}));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dyT[r] = f2h(t0[r]);
        dyT[4 + r] = f2h(t1[r]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const h8 sel = (nt & 1) ? I1 : I0;
      f4 t0 = MFMA(h2[0][nt >> 1], sel, (f4{0, 0, 0, 0}));
      f4 t1 = MFMA(h2[1][nt >> 1], sel, (f4{0, 0, 0, 0}));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hT[nt][r] = f2h(t0[r]);
        hT[nt][4 + r] = f2h(t1[r]);
      }
      dWo0[nt] += MFMA(dyT, hT[nt], (f4{0, 0, 0, 0}))[0];
    }
    {
      // (orientation 2 first -- it needs the old chain fragments --, then the chain fragments in place, two C tiles at a time: the
      // order in which the registers die decides whether the kernel fits two wavefronts per SIMD)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f4 t0 = MFMA(dzf[0][0], FR(ATTR_BF_WOT_N + nt), (f4{0, 0, 0, 0}));
        f4 t1 = MFMA(dzf[1][0], FR(ATTR_BF_WOT_N + nt), (f4{0, 0, 0, 0}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dzT[nt][r] = (hT[nt][r] > (half_t)0.0f) ? f2h_grad(t0[r]) : (half_t)0.0f;
          dzT[nt][4 + r] = (hT[nt][4 + r] > (half_t)0.0f) ? f2h_grad(t1[r]) : (half_t)0.0f;
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const h8 dy8 = dzf[q][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f4 c0 = MFMA(FR(ATTR_BF_WOT_P + 2 * ks), dy8, (f4{0, 0, 0, 0}));
          f4 c1 = MFMA(FR(ATTR_BF_WOT_P + 2 * ks + 1), dy8, (f4{0, 0, 0, 0}));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dzf[q][ks][r] = (h2[q][ks][r] > (half_t)0.0f) ? f2h_grad(c0[r]) : (half_t)0.0f;
            dzf[q][ks][4 + r] = (h2[q][ks][4 + r] > (half_t)0.0f) ? f2h_grad(c1[r]) : (half_t)0.0f;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (nothing of the next step -- its LDS fragment reads above all -- is scheduled up into this one: registers)
    // ---- hidden layer 2: dW2 = dZ2^T H1, dZ1 ----
    {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const h8 sel = (nt & 1) ? I1 : I0;
        f4 t0 = MFMA(h1[0][nt >> 1], sel, (f4{0, 0, 0, 0}));
        f4 t1 = MFMA(h1[1][nt >> 1], sel, (f4{0, 0, 0, 0}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hT[nt][r] = f2h(t0[r]);
          hT[nt][4 + r] = f2h(t1[r]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dW2[mt][nt] = MFMA(dzT[mt], hT[nt], dW2[mt][nt]);
      {
        h8 nzT[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f4 t0 = MFMA(dzf[0][0], FR(ATTR_BF_W2T + 8 + nt * 2 + 0), (f4{0, 0, 0, 0}));
          t0 = MFMA(dzf[0][1], FR(ATTR_BF_W2T + 8 + nt * 2 + 1), t0);
          f4 t1 = MFMA(dzf[1][0], FR(ATTR_BF_W2T + 8 + nt * 2 + 0), (f4{0, 0, 0, 0}));
          t1 = MFMA(dzf[1][1], FR(ATTR_BF_W2T + 8 + nt * 2 + 1), t1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            nzT[nt][r] = (hT[nt][r] > (half_t)0.0f) ? f2h_grad(t0[r]) : (half_t)0.0f;
            nzT[nt][4 + r] = (hT[nt][4 + r] > (half_t)0.0f) ? f2h_grad(t1[r]) : (half_t)0.0f;
          }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) dzT[nt] = nzT[nt];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const h8 d0 = dzf[q][0], d1 = dzf[q][1];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f4 c0 = MFMA(FR(ATTR_BF_W2T + (2 * ks) * 2 + 0), d0, (f4{0, 0, 0, 0}));
          c0 = MFMA(FR(ATTR_BF_W2T + (2 * ks) * 2 + 1), d1, c0);
          f4 c1 = MFMA(FR(ATTR_BF_W2T + (2 * ks + 1) * 2 + 0), d0, (f4{0, 0, 0, 0}));
          c1 = MFMA(FR(ATTR_BF_W2T + (2 * ks + 1) * 2 + 1), d1, c1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dzf[q][ks][r] = (h1[q][ks][r] > (half_t)0.0f) ? f2h_grad(c0[r]) : (half_t)0.0f;
            dzf[q][ks][4 + r] = (h1[q][ks][4 + r] > (half_t)0.0f) ? f2h_grad(c1[r]) : (half_t)0.0f;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (nothing of the next step -- its LDS fragment reads above all -- is scheduled up into this one: registers)
    // ---- first layer: dW1 of the 32 per-sample columns; the row sums of dZ1 per ray; dX of the 16 geometry columns ----
    // (two-wavefront build: the rows are fetched a second time here -- an L1 hit -- instead of held across the chain: 8 registers)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      uint4 u = cur.x[q];
      if (!PF) {
        u = *attr_x_ptr(a.h, ps[q], g);
        asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));
      }
      u = attr_x_fix(g, u);
      if (g == 3) {  // columns 96 / 97: the ray indicators (see ROW SUMS above); 98 .. 103: nothing
        const unsigned long long mu = q ? u1 : u0, mv = q ? v1 : v0;
        u = make_uint4((((mu >> lane) & 1ull) ? 0x3C00u : 0u) | (((mv >> lane) & 1ull) ? 0x3C000000u : 0u), 0u, 0u, 0u);
      }
      if (!ok[q]) u = make_uint4(0, 0, 0, 0);
      xf[q] = *reinterpret_cast<h8*>(&u);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const h8 sel = nt ? I1 : I0;
      f4 t0 = MFMA(xf[0], sel, (f4{0, 0, 0, 0}));
      f4 t1 = MFMA(xf[1], sel, (f4{0, 0, 0, 0}));
      h8 xT;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xT[r] = f2h(t0[r]);
        xT[4 + r] = f2h(t1[r]);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) dW1[mt][nt] = MFMA(dzT[mt], xT, dW1[mt][nt]);
    }
    if (ray_v >= 0) {  // both rays leave the registers: the one that ends here, and the one that goes on (it starts again in the next tile)
      flush(8, ray_u);
      flush(9, ray_v);
      acc_ray = -1;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f4 c = MFMA(FR(ATTR_BF_W1T + 0), dzf[q][0], (f4{0, 0, 0, 0}));
      c = MFMA(FR(ATTR_BF_W1T + 1), dzf[q][1], c);
      if (ok[q]) {
        // physical columns 72 .. 87 = [1.0 | g0 .. g14] = the sigma network's output row: this 4-column piece lands at dh[sample][4g]
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = f2h_grad(c[r]);
        const h4 old = *reinterpret_cast<const h4*>(&dh_old[q]);
        if (b.add) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = f2h_grad(h2f(old[r]) + h2f(ov[r]));
        }
        if (g == 0) ov[0] = old[0];  // column 0: the density activation's adjoint stays
        *reinterpret_cast<h4*>(b.dh + (int64_t)ps[q] * 16 + 4 * g) = ov;
      }
    }
    if (PF) cur = nxt;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      ent_cur[q] = ent_nxt[q];
      ent_nxt[q] = ent_nn[q];
    }
  }
  if (acc_ray >= 0) flush(8, acc_ray);
  // ---- flush dW (fp32 atomics; one add per element per wave) ----
  float* gW1 = b.grad_w;
  float* gW2 = b.grad_w + HID * ATTR_IN;
  float* gWo = gW2 + HID * HID;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = ATTR_ENC + 16 * nt + i;  // physical column
        const float v = dW1[mt][nt][r] * b.inv_scale;
        if (c < ATTR_IN && v != 0.0f) atomicAdd(gW1 + (16 * mt + 4 * g + r) * ATTR_IN + col_map(c, ATTR_ENC), v);
      }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = dW2[mt][nt][r] * b.inv_scale;
        if (v != 0.0f) atomicAdd(gW2 + (16 * mt + 4 * g + r) * HID + 16 * nt + i, v);
      }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float v = dWo0[nt] * b.inv_scale;
    if (g == 0 && v != 0.0f) atomicAdd(gWo + 16 * nt + i, v);
  }
}

// ---- dW1[:, 0:72] from the flushed row sums -----------------------------------------------------
// grad_w[n][c] += inv_scale * sum_slots part[slot][n] * enc[ray(slot)][c].  One workgroup walks a range of slots; thread = (neuron n,
// column group q of 18 columns); fp32 throughout.
__global__ void __launch_bounds__(256) attr_enc_grad_kernel(const float* __restrict__ part, const int32_t* __restrict__ part_count,
                                                          int32_t part_cap, const half_t* __restrict__ enc, float inv_scale,
                                                          float* __restrict__ grad_w) {
  const int n = threadIdx.x & 63, q = threadIdx.x >> 6;
  constexpr int CG = ATTR_ENC / 4;  // 18 columns per thread
  const int32_t n_slots = min(*part_count, part_cap);
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.0f;
  // four slots per iteration: their (dependent) loads -- the slot's ray, then that ray's encoding -- are in flight together
  for (int32_t s0 = blockIdx.x * 4; s0 < n_slots; s0 += gridDim.x * 4) {
    float v[4];
    const half_t* er[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int32_t s = min(s0 + u, n_slots - 1);
      const float* p = part + (int64_t)s * ATTR_PART_STRIDE;
      v[u] = s0 + u < n_slots ? p[n] : 0.0f;
      er[u] = enc + (int64_t)(*reinterpret_cast<const int32_t*>(p + 64)) * ATTR_ENC + q * CG;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = fmaf(v[u], h2f(er[u][c]), acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    const float v = acc[c] * inv_scale;
    if (v != 0.0f) atomicAdd(grad_w + n * ATTR_IN + q * CG + c, v);
  }
}

// ================================================================================================
// C ABI
// ================================================================================================
static int attr_nets_check(const char* who, int64_t cap, int32_t n_rays, int32_t T, int32_t n_enc, int32_t n_geo, int32_t in_pad, int32_t n_hidden) {
  if (in_pad != ATTR_IN || n_enc != ATTR_ENC || n_geo != 15 || n_hidden != 2 || T <= 0 || n_rays <= 0 || cap >= ((int64_t)1 << 31)) {
    l4d_set_error(1, who);
    return 1;
  }
  return 0;
}

// room for the flushed row sums of BOTH launches.  A wavefront flushes when the ray it is summing changes and twice more per tile
// that holds two rays: at most three times per tile, whatever the work list looks like.
extern "C" int64_t l4d_attr_nets_bwd_workspace(int32_t n_rays, int64_t cap) {
  (void)n_rays;
  const int64_t slots = 2 * (3 * ((cap + 31) / 32 + 1) + 4096);
  return 256 + slots * ATTR_PART_STRIDE * 4;
}

extern "C" int l4d_attr_nets_fwd(const int32_t* idx, const int32_t* count, int64_t cap, int32_t n_rays, int32_t T, const void* dir_enc,
                                 int32_t n_enc, const void* h, int32_t n_geo, int32_t in_pad, int32_t n_hidden, const void* w_raydrop,
                                 const void* w_intensity, float* ray_term, float* attr_dense, float* attr_compact, void* stream) {
  if (cap == 0) return 0;
  if (attr_nets_check("l4d_attr_nets_fwd: needs in_pad 96, n_enc 72, n_geo 15, n_hidden 2", cap, n_rays, T, n_enc, n_geo, in_pad, n_hidden)) return 1;
  L4D_LAUNCH(attr_ray_term_kernel, dim3((unsigned)std::min<int64_t>((n_rays + 3) / 4, 2048)), dim3(256), 0, (hipStream_t)stream,
             (const half_t*)w_raydrop, (const half_t*)w_intensity, (const half_t*)dir_enc, n_rays, ray_term);
  const AttrNets a{idx, count, (const half_t*)h, ray_term, cap, n_rays, T};
  int64_t blocks = ((cap + 15) / 16 + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  L4D_LAUNCH(attr_nets_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (const half_t*)w_raydrop,
             (const half_t*)w_intensity, attr_dense, attr_compact);
  L4D_LAUNCH_CHECK("l4d_attr_nets_fwd");
  return 0;
}

extern "C" int l4d_attr_nets_bwd(const int32_t* idx, const int32_t* count, int64_t cap, int32_t n_rays, int32_t T, const void* dir_enc,
                                 int32_t n_enc, const void* h, int32_t n_geo, int32_t in_pad, int32_t n_hidden, const void* w_raydrop,
                                 const void* w_intensity, const float* ray_term, const float* d_attr, const float* attr_compact,
                                 float loss_scale, void* dh, float* grad_raydrop, float* grad_intensity, float inv_loss_scale,
                                 void* workspace, void* stream) {
  if (cap == 0) return 0;
  if (attr_nets_check("l4d_attr_nets_bwd: needs in_pad 96, n_enc 72, n_geo 15, n_hidden 2", cap, n_rays, T, n_enc, n_geo, in_pad, n_hidden)) return 1;
  static int n_cu = 0;
  if (n_cu <= 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
  }
  char* ws = (char*)workspace;
  int32_t* part_count = (int32_t*)ws;
  float* part = (float*)(ws + 256);
  const int64_t slots_all = (l4d_attr_nets_bwd_workspace(n_rays, cap) - 256) / (ATTR_PART_STRIDE * 4);
  const int32_t part_cap = (int32_t)(slots_all / 2);
  l4d_fill_async(ws, 0u, 256, (hipStream_t)stream);
  const AttrNets a{idx, count, (const half_t*)h, ray_term, cap, n_rays, T};
  // two workgroups per CU (two wavefronts per SIMD), every wavefront one contiguous range of tiles; the flush of dW at the end is
  // a few thousand atomics per wavefront on the same addresses: no more workgroups than fill the chip
  int64_t blocks = ((cap + 31) / 32 + 3) / 4;
  if (blocks > ATTR_BWD_WAVES * n_cu) blocks = ATTR_BWD_WAVES * n_cu;
  for (int ch = 0; ch < 2; ++ch) {
    const AttrBwd b{d_attr, attr_compact, (half_t*)dh, ch ? grad_intensity : grad_raydrop, part + (int64_t)ch * part_cap * ATTR_PART_STRIDE,
                    part_count + ch, part_cap, ch, ch, loss_scale, inv_loss_scale};
    L4D_LAUNCH(attr_net_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a,
               (const half_t*)(ch ? w_intensity : w_raydrop), b);
    L4D_LAUNCH(attr_enc_grad_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, b.part, b.part_count, part_cap, (const half_t*)dir_enc,
               inv_loss_scale, b.grad_w);
  }
  L4D_LAUNCH_CHECK("l4d_attr_nets_bwd");
  return 0;
}
