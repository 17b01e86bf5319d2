// Fully-fused bias-free ReLU MLP on the gfx950 matrix cores, replacing tcnn.Network(FullyFusedMLP)
// (model/lidar4d.py:83-117: sigma_net, intensity_net, raydrop_net) and the nn.Linear stack of
// model/flow_field.py:84-98.  Hidden width 64, output padded to 16, input padded to a multiple of 16.
//
// Everything between the global loads and stores lives in registers: one wave owns a tile of batch
// rows and chains v_mfma_f32_16x16x32_f16 (fp16 operands, fp32 accumulate) through the layers.
//
//   MFMA 16x16x32 operand layout (lane l: i = l & 15, g = l >> 4):
//     A[i][8g..8g+7]   B[8g..8g+7][i]   C/D[4g + r][i], r = 0..3
//   A and B have the same per-lane shape ("one row/column index, 8 consecutive k"), so a register set
//   can be used as either operand.
//
// Forward orientation ("chain"): batch rows are the N dimension, C[neuron][row].  A lane then holds 4
// neurons of ONE row per output tile; weight rows are loaded in the permuted order
//   n1(mt, i) = 32*(mt>>1) + 8*(i>>2) + 4*(mt&1) + (i&3)
// so that tiles 2ks and 2ks+1 together give the lane neurons 32ks+8g .. 32ks+8g+7 of its row: exactly
// the next layer's B fragment.  No LDS, no shuffles between layers.
//
// Backward needs contractions over the BATCH (dW = dZ^T H).  Those want "8 consecutive rows per lane".
// Instead of transposing through LDS the same quantities are produced a second time with the batch as
// the M dimension ("orientation 2": C'[row][feature], lane = feature, 4 rows per tile), using the
// chain-layout registers as A operand, and raw inputs are transposed with an identity-matrix MFMA
// (exact).  Macro tile = 32 rows = chain tiles a=0,1 with row(a, i) = 8*(i>>2) + 4a + (i&3), which
// makes the two orientation-2 tiles give the lane rows 8g .. 8g+7.
// MFMA utilisation is irrelevant here (~2 % of peak at target rate); the doubled MFMA work
// buys a kernel with no LDS traffic in the loop and no barriers.
#include "common.h"
#include "wave_dev.h"

#include "mlp_dev.h"

// Input rows of the attribute networks assembled on the fly (model/lidar4d.py:196-213: row j of the work list =
// [frequency encoding of the ray direction (n_enc) | geo_feat of sample idx[j] = h[:, 1 : 1 + n_geo] | ones]) instead of
// being materialised as a [rows, in_pad] matrix that is written once and read four times (two networks, forward and backward).
struct AttrSrc {
  const int32_t* idx;   // work list (null: row j is sample j)
  const half_t* denc;   // [rays, n_enc] fp16, n_enc a multiple of 8
  const half_t* h;      // [samples, 16] fp16 sigma-network output: geo_feat = columns 1 .. 15
  int T, n_enc;
  int cperm;            // = n_enc when gathering, -1 otherwise (col_map)
};
// PHYSICAL columns k0 .. k0 + 7 of the row of sample p (k0 a multiple of 8; column order: col_map with cperm = n_enc), in two
// steps so that the load can be issued a tile ahead with nothing touching its destination: attr_chunk_ptr = where the 16
// bytes come from (always a valid address, no branch), attr_chunk_fix = what is patched once they have arrived.
__device__ __forceinline__ const uint4* attr_chunk_ptr(const AttrSrc& s, int64_t p, int k0) {
  const int q = (k0 - s.n_enc) >> 3;
  const half_t* enc = s.denc + (int64_t)((uint32_t)p / (uint32_t)s.T) * s.n_enc + min(k0, s.n_enc - 8);  // p < 2^31 (checked at launch)
  const half_t* geo = s.h + p * 16 + 8 * (q & 1);  // q >= 2: any valid 16 bytes, replaced by ones below
  return reinterpret_cast<const uint4*>(k0 + 8 <= s.n_enc ? enc : geo);
}
__device__ __forceinline__ uint4 attr_chunk_fix(const AttrSrc& s, int k0, uint4 u) {
  const int q = (k0 - s.n_enc) >> 3;
  if (k0 + 8 <= s.n_enc) return u;
  if (q >= 2) return make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);  // ones
  if (q == 0) u.x = (u.x & 0xFFFF0000u) | 0x3C00u;  // [1.0, g0 .. g6]: the sigma logit's slot carries the constant
  return u;
}

// ================================================================================================
// forward
// ================================================================================================
// weight layout (fp16): W1 [64, in_pad], (NH-1) x [64, 64], Wo [16, 64]
// GATHER: input rows assembled from AttrSrc; WRITE_X: the assembled rows (physical column order) are also stored to xout
// [cap, IN_PAD] for a backward pass that reads them as a plain matrix.
// EPI: activation epilogue on output column 0 ({none, trunc_exp@col0, sigmoid}):
//   0  with epi0 given: epi0[row] = exp(y0)     -- the density activation (activation.py:6-20), fp32, y still stored
//   2  s = fp16(sigmoid(y0)); epi0[2 * sample + epi_ch] = s (dense [P, 2] image of lidar4d.py:216-219), epi1[2 * row + epi_ch] = s
//      (compact copy for the backward); y itself is not stored (nothing reads it)
struct MlpEpi {
  float* epi0;
  float* epi1;
  int ch;
};
template <int IN_TILES, int NH, bool GATHER = false, bool WRITE_X = false, int EPI = 0>
__global__ void __launch_bounds__(256) mlp_fwd_kernel(const half_t* __restrict__ x, int64_t cap, const int32_t* __restrict__ n_rows,
                                                     const half_t* __restrict__ weights, half_t* __restrict__ y,
                                                     half_t* __restrict__ act, AttrSrc src, half_t* __restrict__ xout = nullptr,
                                                     MlpEpi epi = MlpEpi{nullptr, nullptr, 0}) {
  // cap = rows the buffers were sized for (stride of the act planes); P = rows actually present
  const int64_t P = n_rows ? min((int64_t)*n_rows, cap) : cap;
  constexpr int IN_PAD = IN_TILES * 16;
  constexpr int KS_IN = (IN_TILES + 1) / 2;
  constexpr int NF_L1 = 4 * KS_IN;
  constexpr int NF = NF_L1 + (NH - 1) * 8 + 2;
  __shared__ uint4 frags[NF][64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;

  // build A fragments cooperatively
  for (int f = wave; f < NF; f += 4) {
    h8 v;
    if (f < NF_L1) {
      const int mt = f / KS_IN, ks = f % KS_IN;
      v = build_frag(weights, HID, IN_PAD, 0, perm_row(mt, i), 32 * ks + 8 * g, src.cperm);
    } else if (f < NF_L1 + (NH - 1) * 8) {
      const int q = f - NF_L1, layer = q / 8, mt = (q % 8) / 2, ks = q % 2;
      v = build_frag(weights + HID * IN_PAD + layer * HID * HID, HID, HID, 0, perm_row(mt, i), 32 * ks + 8 * g);
    } else {
      const int ks = f - (NF_L1 + (NH - 1) * 8);
      v = build_frag(weights + HID * IN_PAD + (NH - 1) * HID * HID, 16, HID, 0, i, 32 * ks + 8 * g);
    }
    frags[f][lane] = *reinterpret_cast<uint4*>(&v);
  }
  __syncthreads();
  auto FR = [&](int f) -> h8 { uint4 u = frags[f][lane]; return *reinterpret_cast<h8*>(&u); };

  const int64_t n_tiles = (P + 15) / 16;
  // PREFETCH: a tile's input rows are fetched one tile ahead and its work-list entry two tiles ahead (the row loads of a
  // gathered tile depend on it): the loads of tile t + 1 fly while tile t is computed.  Tiles / rows past the end re-read
  // valid rows (clamped) and are never stored.
  const int64_t tile_stride = (int64_t)gridDim.x * 4;
  auto tile_row = [&](int64_t tile) -> int64_t {  // this lane's row of `tile`, clamped
    const int64_t row = min(tile, n_tiles - 1) * 16 + i;
    return row < P ? row : 0;
  };
  // work-list entry of the row: loaded without a branch (no work list: any readable word, ignored by tile_src) so that no join
  // makes the compiler wait for it where it is issued
  const int32_t* entries = GATHER ? (src.idx ? src.idx : reinterpret_cast<const int32_t*>(src.h)) : nullptr;
  auto load_entry_of = [&](int64_t tile) -> int32_t { return GATHER ? entries[tile_row(tile)] : 0; };
  auto tile_src = [&](int64_t tile, int32_t entry) -> int64_t {  // source sample of the row
    return (GATHER && src.idx) ? (int64_t)entry : tile_row(tile);
  };
  auto load_rows = [&](int64_t ps, uint4 xr[KS_IN]) {  // ps = tile_src(tile)
#pragma unroll
    for (int ks = 0; ks < KS_IN; ++ks) {
      const int k0 = 32 * ks + 8 * g;
      if (k0 < IN_PAD) xr[ks] = GATHER ? *attr_chunk_ptr(src, ps, k0) : *reinterpret_cast<const uint4*>(x + ps * IN_PAD + k0);
      else xr[ks] = make_uint4(0, 0, 0, 0);
    }
  };
  // (only the gathered variants: a plain row matrix is streamed by four wavefronts per SIMD that hide each other's latency,
  // measured 0.96 -> 1.05 ms with the prefetch; the gathered ones have the dependent work-list load in front: 1.91 -> 1.78)
  constexpr bool PREFETCH = GATHER;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  int64_t ps_cur = 0;
  int32_t entry_nxt = 0;
  uint4 xc[KS_IN];
  if (PREFETCH && tile < n_tiles) {
    ps_cur = tile_src(tile, load_entry_of(tile));
    entry_nxt = load_entry_of(tile + tile_stride);
    load_rows(ps_cur, xc);
  }
  for (; tile < n_tiles; tile += tile_stride) {
    int64_t ps_nxt = 0;
    int32_t entry_nn = 0;
    uint4 xn[KS_IN];
    if (PREFETCH) {
      asm volatile("" ::: "memory");
      ps_nxt = tile_src(tile + tile_stride, entry_nxt);
      load_rows(ps_nxt, xn);
      entry_nn = load_entry_of(tile + 2 * tile_stride);
      asm volatile("" ::: "memory");  // the prefetch stays up here
    } else {
      ps_cur = tile_src(tile, load_entry_of(tile));
      load_rows(ps_cur, xc);
    }
    const int64_t row = tile * 16 + i;
    const bool ok = row < P;
    const int64_t psrc = ps_cur;
    // input fragments
    h8 xb[KS_IN];
#pragma unroll
    for (int ks = 0; ks < KS_IN; ++ks) {
      const int k0 = 32 * ks + 8 * g;
      if (GATHER && k0 < IN_PAD) xc[ks] = attr_chunk_fix(src, k0, xc[ks]);
      xb[ks] = *reinterpret_cast<h8*>(&xc[ks]);
      if (WRITE_X && ok && k0 < IN_PAD) *reinterpret_cast<uint4*>(xout + row * IN_PAD + k0) = xc[ks];
    }
    // layer 1
    f4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      acc[mt] = f4{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) acc[mt] = MFMA(FR(mt * KS_IN + ks), xb[ks], acc[mt]);
    }
    h8 hb[2];
    hb[0] = relu_pack(acc[0], acc[1]);
    hb[1] = relu_pack(acc[2], acc[3]);
    if (act && ok) {
      *reinterpret_cast<h8*>(act + row * HID + 8 * g) = hb[0];
      *reinterpret_cast<h8*>(act + row * HID + 32 + 8 * g) = hb[1];
    }
    // further hidden layers
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      const int base = NF_L1 + (l - 1) * 8;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        acc[mt] = f4{0, 0, 0, 0};
        acc[mt] = MFMA(FR(base + mt * 2 + 0), hb[0], acc[mt]);
        acc[mt] = MFMA(FR(base + mt * 2 + 1), hb[1], acc[mt]);
      }
      hb[0] = relu_pack(acc[0], acc[1]);
      hb[1] = relu_pack(acc[2], acc[3]);
      if (act && ok) {
        half_t* a_l = act + (int64_t)l * cap * HID;
        *reinterpret_cast<h8*>(a_l + row * HID + 8 * g) = hb[0];
        *reinterpret_cast<h8*>(a_l + row * HID + 32 + 8 * g) = hb[1];
      }
    }
    // output layer: C[m = 4g + r][row i]
    f4 o = f4{0, 0, 0, 0};
    o = MFMA(FR(NF - 2), hb[0], o);
    o = MFMA(FR(NF - 1), hb[1], o);
    if (ok) {
      h4 ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = f2h(clamp_h(o[r]));
      if (EPI != 2) *reinterpret_cast<h4*>(y + row * 16 + 4 * g) = ov;
      if (EPI != 2 && epi.epi0 && g == 0) epi.epi0[row] = expf(h2f(ov[0]));
      if (EPI == 2 && g == 0) {
        const float sgm = h2f(f2h(1.0f / (1.0f + expf(-h2f(ov[0])))));
        epi.epi0[psrc * 2 + epi.ch] = sgm;
        epi.epi1[row * 2 + epi.ch] = sgm;
      }
    }
    if (PREFETCH) {
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) xc[ks] = xn[ks];
      ps_cur = ps_nxt;
      entry_nxt = entry_nn;
    }
  }
}

// ================================================================================================
// backward
// ================================================================================================
template <int IN_TILES, int NH>
struct BwdFrags {
  static constexpr int KS_IN = (IN_TILES + 1) / 2;
  // W_o^T : chain (permuted rows) 4, orientation-2 (natural rows) 4
  static constexpr int WOT_P = 0;
  static constexpr int WOT_N = 4;
  // hidden W_l^T, l = NH .. 2: per layer chain 8 + natural 8
  static constexpr int WH = 8;
  // W_1^T natural: IN_TILES x 2
  static constexpr int W1T = WH + (NH - 1) * 16;
  static constexpr int NF = W1T + IN_TILES * 2;
};

// COL_LO..COL_HI: range of 16-wide input-column tiles whose dW1 / dX this launch produces; REST: also accumulate the
// hidden/output layers' dW.  Inputs wider than 128 columns are handled by two launches (the dW1 accumulators of all
// columns do not fit the register file next to the other layers'); the chain is recomputed, which is cheap.
// RECOMP: the hidden activations are not read from `act` but recomputed from x with the forward chain (forward weight
// fragments in LDS as well): trades 128 B/row/layer of HBM traffic (written by the forward, read here) for 4 KS_IN + 8 (NH-1)
// MFMAs per 16 rows.  Used where the activations dominate the traffic (the flow network: 32-byte rows, 256 B of activations).
// GATHER: x rows come from AttrSrc; DX_LO: dx is produced only for the column tiles DX_LO .. COL_HI - 1 and stored compactly
// as [rows, (COL_HI - DX_LO) * 16] (the attribute networks need the gradient of their geo_feat columns only: the direction
// encoding has no trainable input).
// MLP_BWD_NARROW_WAVES: waves per SIMD the 16-wide (flow) network's backward is compiled for.  Its accumulators are small, but
// the compiler keeps the LDS weight fragments in registers across the tile loop as long as it has any (446 of 512).
#define MLP_BWD_PIN 1  // backward kernels: wait for the prefetched next tile in front of the current tile's dX stores (see there)
#define MLP_BWD_NARROW_WAVES 2  // round 5: 248 registers without the next-tile prefetch, two workgroups per CU: 1.03 -> 0.93 ms (gpurun_out/s13; with one
                                // workgroup per CU and no prefetch: 1.28; round 2's two-wave attempt kept the prefetch and spilled 80 bytes)
// ATTR_EPI (attribute networks, with GATHER): the two streaming steps around the network live in the kernel --
//   in:  dy[row][0] = d_attr[sample][ch] * s (1 - s) * loss_scale with s = attr_compact[row][ch] (adjoint of the sigmoid +
//        scatter, lidar4d.py:210-219), other columns 0, instead of a [rows, 16] matrix that is 15/16 zeros;
//   out: the geo-feature columns of dX go to dh[sample][0 .. 15] (the sigma network's output gradient; column 0 is written
//        later by the density activation's adjoint) -- stored by the first network, added by the second -- instead of two
//        [rows, 32] matrices and a gather kernel that sums them.
struct AttrBwdEpi {
  const float* d_attr;   // [samples, 2]
  const float* attr_c;   // [rows, 2]
  half_t* dh;            // [samples, 16]
  int ch, accumulate;    // accumulate: bit 0 = add to what dh holds (else store), bit 1 = dh column 0 already carries the density
  float loss_scale;      // activation's adjoint (l4d_sigma_bwd_rows ran first) and is kept instead of being zeroed
};
// DxStat: *out = max(*out, max |dx[:, 16 lo_tile : 16 hi_tile]|) over the rows of the launch, as stored (fp16-rounded; +inf for a
// non-finite value): what the consumer of those columns needs to scale its fixed-point accumulators (field_bwd.hip).
struct DxStat {
  float* out;
  int lo_tile, hi_tile;
};
template <int IN_TILES, int NH, int COL_LO, int COL_HI, bool REST, bool RECOMP = false, bool GATHER = false, int DX_LO = COL_LO,
          bool ATTR_EPI = false>
__global__ void __launch_bounds__(256, (IN_TILES == 1 && NH <= 2 ? MLP_BWD_NARROW_WAVES : 1)) mlp_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ act,
                                                     const half_t* __restrict__ dy, int64_t cap,
                                                     const int32_t* __restrict__ n_rows,
                                                     const half_t* __restrict__ weights, half_t* __restrict__ dx,
                                                     float* __restrict__ grad_w, float inv_scale, AttrSrc src,
                                                     AttrBwdEpi aepi = AttrBwdEpi{nullptr, nullptr, nullptr, 0, 0, 0.0f},
                                                     DxStat dstat = DxStat{nullptr, 0, 0}) {
  const int64_t P = n_rows ? min((int64_t)*n_rows, cap) : cap;
  using L = BwdFrags<IN_TILES, NH>;
  float dx_amax = 0.0f;  // DxStat: running max |dx| of the requested column tiles (this lane's share)
  constexpr int IN_PAD = IN_TILES * 16;
  constexpr int KS_IN = L::KS_IN;
  constexpr int NF_FWD = RECOMP ? 4 * KS_IN + (NH - 1) * 8 : 0;  // forward chain fragments (layers 1 .. NH), as in mlp_fwd_kernel
  __shared__ uint4 frags[L::NF + NF_FWD][64];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const half_t* W1 = weights;
  const half_t* Wo = weights + HID * IN_PAD + (NH - 1) * HID * HID;

  for (int f = wave; f < L::NF; f += 4) {
    h8 v;
    if (f < L::WH) {
      const int mt = f & 3;
      const int row = (f < 4) ? perm_row(mt, i) : 16 * mt + i;
      v = build_frag(Wo, 16, HID, 1, row, 8 * g);  // k = output index m (only 16 real)
    } else if (f < L::W1T) {
      const int q = f - L::WH, li = q / 16, r = q % 16;  // li = 0 -> layer NH, 1 -> NH-1 ...
      const int layer = NH - li;                         // hidden layer index (>= 2)
      const half_t* Wl = weights + HID * IN_PAD + (layer - 2) * HID * HID;
      const int nat = r / 8, mt = (r % 8) / 2, ks = r % 2;
      const int row = nat ? 16 * mt + i : perm_row(mt, i);
      v = build_frag(Wl, HID, HID, 1, row, 32 * ks + 8 * g);
    } else {
      const int q = f - L::W1T, mt = q / 2, ks = q % 2;
      v = build_frag(W1, HID, IN_PAD, 1, 16 * mt + i, 32 * ks + 8 * g, src.cperm);
    }
    frags[f][lane] = *reinterpret_cast<uint4*>(&v);
  }
  if (RECOMP) {
    for (int f = wave; f < NF_FWD; f += 4) {
      h8 v;
      if (f < 4 * KS_IN) {
        const int mt = f / KS_IN, ks = f % KS_IN;
        v = build_frag(weights, HID, IN_PAD, 0, perm_row(mt, i), 32 * ks + 8 * g, src.cperm);
      } else {
        const int q = f - 4 * KS_IN, layer = q / 8, mt = (q % 8) / 2, ks = q % 2;
        v = build_frag(weights + HID * IN_PAD + layer * HID * HID, HID, HID, 0, perm_row(mt, i), 32 * ks + 8 * g);
      }
      frags[L::NF + f][lane] = *reinterpret_cast<uint4*>(&v);
    }
  }
  __syncthreads();
  auto FR = [&](int f) -> h8 { uint4 u = frags[f][lane]; return *reinterpret_cast<h8*>(&u); };
  const h8 I0 = ident_frag(lane, 0), I1 = ident_frag(lane, 1);

  // dW accumulators, C layout: [m = 16*mt + 4g + r][k = 16*nt + i]
  f4 dWo[4];
  f4 dWh[(NH > 1 ? NH - 1 : 1)][4][4];
  constexpr int NCOL = COL_HI - COL_LO;
  f4 dW1[4][NCOL];
#pragma unroll
  for (int a = 0; a < 4; ++a) dWo[a] = f4{0, 0, 0, 0};
#pragma unroll
  for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dWh[l][a][b] = f4{0, 0, 0, 0};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NCOL; ++b) dW1[a][b] = f4{0, 0, 0, 0};

  const int64_t n_macro = (P + 31) / 32;
  // Inputs of one macro tile (32 rows): x rows, dy rows and -- unless recomputed -- the hidden activations, as loaded.  They
  // are fetched ONE TILE AHEAD: with the single wavefront per SIMD this kernel's accumulators leave room for, nothing else
  // hides the HBM latency of a tile's loads (measured: 12,000 clocks per tile for 5,000 clocks of MFMA + VALU work).
  constexpr int NACT = RECOMP ? 0 : NH;
  // (the widest 3-hidden-layer variant has no register left for it; the 16-wide network at two wavefronts per SIMD hides a tile's
  // loads behind its partner wavefront instead: with the prefetch it does not fit 256 registers)
  constexpr bool PREFETCH = !(NH >= 3 && COL_HI - COL_LO >= 8) && !(IN_TILES == 1 && NH <= 2 && MLP_BWD_NARROW_WAVES >= 2);
  struct TileIn {
    uint4 x[2][KS_IN];
    uint4 dy[2];
    uint4 h[NACT > 0 ? NACT : 1][2][2];
    uint2 dh_old[2];  // (ATTR_EPI, accumulate) this lane's 4-column piece of dh[sample]: what the other network stored
  };
  // load_tile only ISSUES loads (no instruction may touch the destination registers before the tile is consumed, or the
  // compiler has to wait for the data right here); finish_tile zeroes what does not exist, one iteration later.
  // (GATHER) the work-list entries of a tile's rows are loaded a further tile ahead: the row loads depend on them
  const int32_t* entries = GATHER ? (src.idx ? src.idx : reinterpret_cast<const int32_t*>(src.h)) : nullptr;
  auto tile_row = [&](int64_t mtile, int a) -> int64_t {
    const int64_t row = min(mtile, n_macro - 1) * 32 + 8 * (i >> 2) + 4 * a + (i & 3);
    return row < P ? row : 0;  // rows past the end read row 0 and are zeroed by finish_tile (P > 0 inside the loop)
  };
  auto load_entries = [&](int64_t mtile, int32_t e[2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a) e[a] = GATHER ? entries[tile_row(mtile, a)] : 0;
  };
  auto load_tile = [&](int64_t mtile, const int32_t e[2], TileIn& t) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int64_t rc = tile_row(mtile, a);
      const int64_t ps = (GATHER && src.idx) ? (int64_t)e[a] : rc;
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) {
        const int k0 = 32 * ks + 8 * g;
        if (k0 < IN_PAD) t.x[a][ks] = GATHER ? *attr_chunk_ptr(src, ps, k0) : *reinterpret_cast<const uint4*>(x + rc * IN_PAD + k0);
        else t.x[a][ks] = make_uint4(0, 0, 0, 0);
      }
      if (ATTR_EPI) {  // the two factors of dy[row][0], as loaded; finish_tile multiplies them
        t.dy[a].x = __float_as_uint(aepi.d_attr[ps * 2 + aepi.ch]);
        t.dy[a].y = __float_as_uint(aepi.attr_c[rc * 2 + aepi.ch]);
        // of the 4-column pieces 16 mt + 4 g of dX exactly one per lane falls into columns n_enc .. n_enc + 15
        // (keep-column-0 mode of a storing launch: only the lanes whose piece starts at column 0 need what dh holds)
        if ((aepi.accumulate & 1) || ((aepi.accumulate & 2) && ((4 * g - src.n_enc) & 15) == 0))
          t.dh_old[a] = *reinterpret_cast<const uint2*>(aepi.dh + ps * 16 + ((4 * g - src.n_enc) & 15));
      } else {
        t.dy[a] = *reinterpret_cast<const uint4*>(dy + rc * 16 + 8 * (g & 1));
      }
#pragma unroll
      for (int l = 0; l < NACT; ++l)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          t.h[l][a][ks] = *reinterpret_cast<const uint4*>(act + (int64_t)l * cap * HID + rc * HID + 32 * ks + 8 * g);
    }
  };
  auto finish_tile = [&](int64_t mtile, TileIn& t) {
    const bool partial = (mtile + 1) * 32 > P;  // wave-uniform: only the last tile
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const bool okr = mtile * 32 + 8 * (i >> 2) + 4 * a + (i & 3) < P;
      if (ATTR_EPI) {
        const float sg = __uint_as_float(t.dy[a].y);
        const half_t v = f2h_grad(__uint_as_float(t.dy[a].x) * sg * (1.0f - sg) * aepi.loss_scale);
        t.dy[a] = make_uint4(g == 0 ? (uint32_t)__builtin_bit_cast(unsigned short, v) : 0u, 0u, 0u, 0u);
      }
      if (g >= 2 || (partial && !okr)) t.dy[a] = make_uint4(0, 0, 0, 0);  // k = 16 .. 31 of the output contraction do not exist
      if (GATHER) {
#pragma unroll
        for (int ks = 0; ks < KS_IN; ++ks)
          if (32 * ks + 8 * g < IN_PAD) t.x[a][ks] = attr_chunk_fix(src, 32 * ks + 8 * g, t.x[a][ks]);
      }
      if (partial) {  // (a real, wave-uniform branch -- the empty asm keeps it one: as selects this was 40 v_cndmask in EVERY tile of the attribute backward)
        asm volatile("" ::: "memory");
        if (!okr) {
#pragma unroll
          for (int ks = 0; ks < KS_IN; ++ks) t.x[a][ks] = make_uint4(0, 0, 0, 0);
#pragma unroll
          for (int l = 0; l < NACT; ++l)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) t.h[l][a][ks] = make_uint4(0, 0, 0, 0);
        }
      }
    }
  };
  // pin_tile: an empty asm per register group of a prefetched tile = its first "use", so that the wait for its loads stands where
  // the caller puts it.  The next tile (requested at the head of an iteration) is taken over in front of the current tile's dX
  // stores, where nothing else is outstanding: left to its first real use -- the head of the next iteration -- the wait drained
  // the stores issued just before it, and in the narrow (flow) network's kernel the compiler, seeing loads pending on the loop's
  // entry edge, waited for part of a tile right behind its own request.  (The first tile is pinned before the loop for that.)
  // Only where dX is at most two column tiles (the attribute networks' 16 geometry columns, the flow network's 16 inputs): their
  // stores sit at the very end of an iteration.  The sigma network stores 8 tiles over the last third of its iteration; pinned in
  // front of the first of them its next tile would have had 160 instructions to arrive -- it keeps the wait at the loop head.
  constexpr bool PIN_NEXT = PREFETCH && MLP_BWD_PIN && (COL_HI - DX_LO) <= 2;
  auto pin_tile = [&](TileIn& t, int32_t (&e)[2]) {
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    typedef uint32_t U2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int ks = 0; ks < KS_IN; ++ks) asm volatile("" : "+v"(reinterpret_cast<U4&>(t.x[a][ks])));
      asm volatile("" : "+v"(reinterpret_cast<U4&>(t.dy[a])));
#pragma unroll
      for (int l = 0; l < NACT; ++l)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(reinterpret_cast<U4&>(t.h[l][a][ks])));
      if (ATTR_EPI) asm volatile("" : "+v"(reinterpret_cast<U2&>(t.dh_old[a])));
      if (GATHER) asm volatile("" : "+v"(e[a]));
    }
  };
  const int64_t tile_stride = (int64_t)gridDim.x * 4;
  int64_t mtile = (int64_t)blockIdx.x * 4 + wave;
  TileIn cur;
  int32_t ent_cur[2] = {0, 0}, ent_nxt[2] = {0, 0};  // work-list entries of the rows in `cur` / of the next tile's rows
  if (PREFETCH && mtile < n_macro) {
    load_entries(mtile, ent_cur);
    load_entries(mtile + tile_stride, ent_nxt);
    load_tile(mtile, ent_cur, cur);
  }
  if (PIN_NEXT) {
    pin_tile(cur, ent_nxt);
    if (GATHER) asm volatile("" : "+v"(ent_cur[0]), "+v"(ent_cur[1]));
  }
  for (; mtile < n_macro; mtile += tile_stride) {
    // the weight fragments are re-read from LDS in every tile: hoisted out of the loop they end up parked in AGPRs and cost 4
    // v_accvgpr_read per use instead of one ds_read_b128
    asm volatile("" ::: "memory");
    TileIn nxt;
    int32_t ent_nn[2] = {0, 0};
    if (PREFETCH) {
      finish_tile(mtile, cur);
      load_tile(mtile + tile_stride, ent_nxt, nxt);  // past the end: the last tile once more, never used
      load_entries(mtile + 2 * tile_stride, ent_nn);
    } else {
      load_entries(mtile, ent_cur);
      load_tile(mtile, ent_cur, cur);
      finish_tile(mtile, cur);
    }
    asm volatile("" ::: "memory");  // keeps the prefetch up here: the scheduler may not sink the loads to their use below
    int64_t rows[2];
    bool ok[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      rows[a] = mtile * 32 + 8 * (i >> 2) + 4 * a + (i & 3);
      ok[a] = rows[a] < P;
    }
    // ---- (RECOMP) forward chain from x: hidden activations of every layer, chain layout [layer][a][ks] ----
    h8 xf[2][KS_IN];
    h8 hrec[RECOMP ? NH : 1][2][2];
    if (RECOMP) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int ks = 0; ks < KS_IN; ++ks) {
          xf[a][ks] = *reinterpret_cast<h8*>(&cur.x[a][ks]);
        }
        f4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          acc[mt] = f4{0, 0, 0, 0};
#pragma unroll
          for (int ks = 0; ks < KS_IN; ++ks) acc[mt] = MFMA(FR(L::NF + mt * KS_IN + ks), xf[a][ks], acc[mt]);
        }
        hrec[0][a][0] = relu_pack(acc[0], acc[1]);
        hrec[0][a][1] = relu_pack(acc[2], acc[3]);
#pragma unroll
        for (int l = 1; l < NH; ++l) {
          const int base = L::NF + 4 * KS_IN + (l - 1) * 8;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            acc[mt] = f4{0, 0, 0, 0};
            acc[mt] = MFMA(FR(base + mt * 2 + 0), hrec[l - 1][a][0], acc[mt]);
            acc[mt] = MFMA(FR(base + mt * 2 + 1), hrec[l - 1][a][1], acc[mt]);
          }
          hrec[l][a][0] = relu_pack(acc[0], acc[1]);
          hrec[l][a][1] = relu_pack(acc[2], acc[3]);
        }
      }
    }
    // ---- output layer ----------------------------------------------------------------------
    h8 dzf[2][2];  // chain B fragments of the current layer's dZ: [a][ks]
    h8 dzT[4];     // orientation-2: lane = feature 16*nt + i, 8 rows
    h8 dyT;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      dzf[a][0] = *reinterpret_cast<h8*>(&cur.dy[a]);
    }
    {
      f4 t0 = MFMA(dzf[0][0], I0, (f4{0, 0, 0, 0}));
      f4 t1 = MFMA(dzf[1][0], I0, (f4{0, 0, 0, 0}));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dyT[r] = f2h(t0[r]);
        dyT[4 + r] = f2h(t1[r]);
      }
    }
    // activations of the last hidden layer, chain layout [a][ks]
    h8 hf[2][2];
    if (RECOMP) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) hf[a][ks] = hrec[NH - 1][a][ks];
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) hf[a][ks] = *reinterpret_cast<h8*>(&cur.h[NACT > 0 ? NH - 1 : 0][a][ks]);
    }
    h8 hT[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const h8 sel = (nt & 1) ? I1 : I0;
      f4 t0 = MFMA(hf[0][nt >> 1], sel, (f4{0, 0, 0, 0}));
      f4 t1 = MFMA(hf[1][nt >> 1], sel, (f4{0, 0, 0, 0}));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hT[nt][r] = f2h(t0[r]);
        hT[nt][4 + r] = f2h(t1[r]);
      }
      if (REST) dWo[nt] = MFMA(dyT, hT[nt], dWo[nt]);
    }
    // dH_NH (chain) and dH_NH' (orientation 2), masked by ReLU
    {
      h8 nz[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f4 c[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          c[mt] = MFMA(FR(L::WOT_P + mt), dzf[a][0], (f4{0, 0, 0, 0}));
          if (!RELU_GATE_ASM) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!(hf[a][mt >> 1][4 * (mt & 1) + r] > (half_t)0.0f)) c[mt][r] = 0.0f;
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (RELU_GATE_ASM) {
            nz[a][ks] = relu_gate(c[2 * ks], c[2 * ks + 1], hf[a][ks]);  // (element 4 (mt & 1) + r of hf[a][mt >> 1] gates c[mt][r])
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              nz[a][ks][r] = f2h_grad(c[2 * ks][r]);
              nz[a][ks][4 + r] = f2h_grad(c[2 * ks + 1][r]);
            }
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f4 t0 = MFMA(dzf[0][0], FR(L::WOT_N + nt), (f4{0, 0, 0, 0}));
        f4 t1 = MFMA(dzf[1][0], FR(L::WOT_N + nt), (f4{0, 0, 0, 0}));
        if (RELU_GATE_ASM) {
          dzT[nt] = relu_gate(t0, t1, hT[nt]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dzT[nt][r] = (hT[nt][r] > (half_t)0.0f) ? f2h_grad(t0[r]) : (half_t)0.0f;
            dzT[nt][4 + r] = (hT[nt][4 + r] > (half_t)0.0f) ? f2h_grad(t1[r]) : (half_t)0.0f;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dzf[a][ks] = nz[a][ks];
    }
    // ---- hidden layers NH .. 2 -------------------------------------------------------------
#pragma unroll
    for (int li = 0; li < NH - 1; ++li) {
      const int layer = NH - li;  // current dZ belongs to hidden layer `layer`; its input is H_{layer-1}
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (RECOMP) {
            hf[a][ks] = hrec[RECOMP ? layer - 2 : 0][a][ks];
          } else {
            hf[a][ks] = *reinterpret_cast<h8*>(&cur.h[NACT > 0 ? layer - 2 : 0][a][ks]);
          }
        }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const h8 sel = (nt & 1) ? I1 : I0;
        f4 t0 = MFMA(hf[0][nt >> 1], sel, (f4{0, 0, 0, 0}));
        f4 t1 = MFMA(hf[1][nt >> 1], sel, (f4{0, 0, 0, 0}));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hT[nt][r] = f2h(t0[r]);
          hT[nt][4 + r] = f2h(t1[r]);
        }
      }
      if (REST) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) dWh[layer - 2][mt][nt] = MFMA(dzT[mt], hT[nt], dWh[layer - 2][mt][nt]);
      }
      const int fb = L::WH + li * 16;
      h8 nz[2][2], nzT[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f4 c[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          c[mt] = MFMA(FR(fb + mt * 2 + 0), dzf[a][0], (f4{0, 0, 0, 0}));
          c[mt] = MFMA(FR(fb + mt * 2 + 1), dzf[a][1], c[mt]);
          if (!RELU_GATE_ASM) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!(hf[a][mt >> 1][4 * (mt & 1) + r] > (half_t)0.0f)) c[mt][r] = 0.0f;
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (RELU_GATE_ASM) {
            nz[a][ks] = relu_gate(c[2 * ks], c[2 * ks + 1], hf[a][ks]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              nz[a][ks][r] = f2h_grad(c[2 * ks][r]);
              nz[a][ks][4 + r] = f2h_grad(c[2 * ks + 1][r]);
            }
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f4 t0 = MFMA(dzf[0][0], FR(fb + 8 + nt * 2 + 0), (f4{0, 0, 0, 0}));
        t0 = MFMA(dzf[0][1], FR(fb + 8 + nt * 2 + 1), t0);
        f4 t1 = MFMA(dzf[1][0], FR(fb + 8 + nt * 2 + 0), (f4{0, 0, 0, 0}));
        t1 = MFMA(dzf[1][1], FR(fb + 8 + nt * 2 + 1), t1);
        if (RELU_GATE_ASM) {
          nzT[nt] = relu_gate(t0, t1, hT[nt]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            nzT[nt][r] = (hT[nt][r] > (half_t)0.0f) ? f2h_grad(t0[r]) : (half_t)0.0f;
            nzT[nt][4 + r] = (hT[nt][4 + r] > (half_t)0.0f) ? f2h_grad(t1[r]) : (half_t)0.0f;
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) dzf[a][ks] = nz[a][ks];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dzT[nt] = nzT[nt];
    }
    // ---- first layer: dW1 = dZ1^T X, dX = dZ1 W1 ---------------------------------------------
    {
      if (!RECOMP) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int ks = 0; ks < KS_IN; ++ks) xf[a][ks] = *reinterpret_cast<h8*>(&cur.x[a][ks]);
      }
#pragma unroll
      for (int nt = COL_LO; nt < COL_HI; ++nt) {
        const h8 sel = (nt & 1) ? I1 : I0;
        f4 t0 = MFMA(xf[0][nt >> 1], sel, (f4{0, 0, 0, 0}));
        f4 t1 = MFMA(xf[1][nt >> 1], sel, (f4{0, 0, 0, 0}));
        h8 xT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xT[r] = f2h(t0[r]);
          xT[4 + r] = f2h(t1[r]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dW1[mt][nt - COL_LO] = MFMA(dzT[mt], xT, dW1[mt][nt - COL_LO]);
      }
      if (PIN_NEXT) pin_tile(nxt, ent_nn);  // the NEXT tile's inputs are taken over here, in front of this tile's dX stores (see pin_tile)
      if (dx || ATTR_EPI) {
        constexpr int DX_PITCH = DX_LO == COL_LO ? IN_PAD : (COL_HI - DX_LO) * 16;  // full rows, or only the tiles from DX_LO on
        constexpr int DX_T0 = DX_LO == COL_LO ? 0 : DX_LO;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int mt = DX_LO; mt < COL_HI; ++mt) {
            f4 c = MFMA(FR(L::W1T + mt * 2 + 0), dzf[a][0], (f4{0, 0, 0, 0}));
            c = MFMA(FR(L::W1T + mt * 2 + 1), dzf[a][1], c);
            if (ok[a]) {
              h4 ov;
#pragma unroll
              for (int r = 0; r < 4; ++r) ov[r] = f2h_grad(c[r]);
              if (ATTR_EPI) {
                // physical columns n_enc .. n_enc + 15 = [1.0 | g0 .. g14] = the sigma network's output row: this 4-column piece
                // lands at dh[sample][piece - n_enc] (column 0, the constant's slot, stays 0)
                const int c0 = 16 * mt + 4 * g - src.n_enc;
                if (c0 >= 0 && c0 < 16) {
                  const int64_t ps = src.idx ? (int64_t)ent_cur[a] : rows[a];
                  h4* d = reinterpret_cast<h4*>(aepi.dh + ps * 16 + c0);
                  const h4 old = *reinterpret_cast<const h4*>(&cur.dh_old[a]);  // fetched with the tile (c0 == (4 g - n_enc) & 15)
                  if (aepi.accumulate & 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = f2h_grad(h2f(old[r]) + h2f(ov[r]));
                  }
                  if (c0 == 0) ov[0] = (aepi.accumulate & 2) ? old[0] : (half_t)0.0f;
                  *d = ov;
                }
              } else {
                *reinterpret_cast<h4*>(dx + rows[a] * DX_PITCH + 16 * (mt - DX_T0) + 4 * g) = ov;
                if (!GATHER && dstat.out && mt >= dstat.lo_tile && mt < dstat.hi_tile) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) dx_amax = amax_nf(dx_amax, h2f(ov[r]));
                }
              }
            }
          }
      }
    }
    if (PREFETCH) {
      cur = nxt;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        ent_cur[a] = ent_nxt[a];
        ent_nxt[a] = ent_nn[a];
      }
    }
  }

  if (!GATHER && dstat.out) {  // wave-uniform
    dx_amax = wave_max(dx_amax);
    if (lane == 0 && dx_amax > 0.0f) atomic_max_nonneg(dstat.out, dx_amax);
  }
  // ---- flush dW (fp32 atomics; one add per element per wave) ----------------------------------
  float* gW1 = grad_w;
  float* gWo = grad_w + HID * IN_PAD + (NH - 1) * HID * HID;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = COL_LO; nt < COL_HI; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = dW1[mt][nt - COL_LO][r] * inv_scale;
        if (v != 0.0f) atomicAdd(gW1 + (16 * mt + 4 * g + r) * IN_PAD + col_map(16 * nt + i, src.cperm), v);
      }
  if (!REST) return;
#pragma unroll
  for (int l = 0; l < NH - 1; ++l) {
    float* gWl = grad_w + HID * IN_PAD + l * HID * HID;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = dWh[l][mt][nt][r] * inv_scale;
          if (v != 0.0f) atomicAdd(gWl + (16 * mt + 4 * g + r) * HID + 16 * nt + i, v);
        }
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = dWo[nt][r] * inv_scale;
      if (v != 0.0f) atomicAdd(gWo + (4 * g + r) * HID + 16 * nt + i, v);
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
// Workgroups of a backward launch: every wavefront ends with a flush of its whole dW (atomics on the same few thousand
// addresses from every wavefront of the grid), so no more workgroups than fill the chip: one per CU where the accumulators
// leave one wavefront per SIMD (everything but the narrowest single-hidden-layer networks), two otherwise.  Measured at C3
// (256 CUs): 256 workgroups 5.57 ms of backward kernels, 512 5.69, 384 7.17 (a round and a half).
static int bwd_grid_cap(int in_pad, int n_hidden) {
  static int n_cu = 0;
  if (n_cu <= 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
      n_cu = 256;
  }
  if (in_pad <= 16 && n_hidden <= 2) return 2 * n_cu;  // compiled for two wavefronts per SIMD (MLP_BWD_NARROW_WAVES): two workgroups per CU
  return (n_hidden >= 2 || in_pad >= 64) ? n_cu : 2 * n_cu;
}

static int grid_for(int64_t tiles) {
  int64_t blocks = (tiles + 3) / 4;
  if (blocks > 2048) blocks = 2048;  // 256 CUs x 8; waves grid-stride over the rest
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

#define FOR_EACH_CFG(X) X(1, 1) X(1, 2) X(1, 3) X(2, 1) X(2, 2) X(2, 3) X(4, 1) X(4, 2) X(4, 3) X(6, 1) X(6, 2) X(6, 3) X(8, 1) X(8, 2) X(8, 3)
#define FOR_EACH_WIDE_CFG(X) X(10, 1) X(10, 2) X(10, 3) X(11, 1) X(11, 2) X(11, 3) X(12, 1) X(12, 2) X(12, 3)

static int mlp_fwd_dispatch(const char* who, const void* x, int64_t P, const int32_t* n_rows, int32_t in_pad, int32_t n_hidden,
                            const void* weights, void* y, void* act, float* sigma, void* stream) {
  if (P == 0) return 0;
  const int in_tiles = in_pad / 16;
  const int grid = grid_for((P + 15) / 16);
  bool done = false;
#define X(IT, NHH)                                                                                                            \
  if (!done && in_pad % 16 == 0 && in_tiles == IT && n_hidden == NHH) {                                                       \
    L4D_LAUNCH((mlp_fwd_kernel<IT, NHH>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, P, n_rows,         \
               (const half_t*)weights, (half_t*)y, (half_t*)act, AttrSrc{nullptr, nullptr, nullptr, 1, 0, -1}, (half_t*)nullptr, \
               MlpEpi{sigma, nullptr, 0});                                                                                    \
    done = true;                                                                                                              \
  }
  FOR_EACH_CFG(X)
  FOR_EACH_WIDE_CFG(X)
#undef X
  if (!done) {
    l4d_set_error(1, "l4d_mlp_fwd: unsupported (in_pad, n_hidden); in_pad in {16,32,64,96,128,160,176,192}, n_hidden in 1..3");
    return 1;
  }
  L4D_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int l4d_mlp_fwd(const void* x, int64_t P, const int32_t* n_rows, int32_t in_pad, int32_t n_hidden,
                           const void* weights, void* y, void* act, void* stream) {
  return mlp_fwd_dispatch("l4d_mlp_fwd", x, P, n_rows, in_pad, n_hidden, weights, y, act, nullptr, stream);
}

// l4d_mlp_fwd with the density activation as epilogue: sigma[row] = exp(y[row][0]) (trunc_exp's forward, activation.py:6-20)
extern "C" int l4d_mlp_fwd_sigma(const void* x, int64_t P, int32_t in_pad, int32_t n_hidden, const void* weights, void* y,
                                 void* act, float* sigma, void* stream) {
  if (!sigma) {
    l4d_set_error(1, "l4d_mlp_fwd_sigma: sigma is null");
    return 1;
  }
  return mlp_fwd_dispatch("l4d_mlp_fwd_sigma", x, P, nullptr, in_pad, n_hidden, weights, y, act, sigma, stream);
}

extern "C" int l4d_mlp_bwd(const void* x, const void* act, const void* dy, int64_t P, const int32_t* n_rows, int32_t in_pad,
                           int32_t n_hidden, const void* weights, void* dx, float* grad_w, float inv_loss_scale,
                           float* dx_absmax, int32_t absmax_col_lo, int32_t absmax_col_hi, void* stream) {
  if (P == 0) return 0;
  if (dx_absmax && (absmax_col_lo % 16 || absmax_col_hi % 16 || absmax_col_lo < 0 || absmax_col_hi > in_pad || !dx)) {
    l4d_set_error(1, "l4d_mlp_bwd: dx_absmax needs dx and a column range in multiples of 16 inside [0, in_pad]");
    return 1;
  }
  const DxStat dstat{dx_absmax, absmax_col_lo / 16, absmax_col_hi / 16};
  const AttrBwdEpi no_epi{nullptr, nullptr, nullptr, 0, 0, 0.0f};
  const int in_tiles = in_pad / 16;
  int grid = grid_for((P + 31) / 32);
  if (grid > bwd_grid_cap(in_pad, n_hidden)) grid = bwd_grid_cap(in_pad, n_hidden);
  bool done = false;
  // act == null: the hidden activations are recomputed from x inside the kernel (narrow inputs only: the flow network)
#define X(IT, NHH)                                                                                                   \
  if (!done && !act && in_pad % 16 == 0 && in_tiles == IT && n_hidden == NHH) {                                      \
    L4D_LAUNCH((mlp_bwd_kernel<IT, NHH, 0, IT, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,          \
               (const half_t*)x, (const half_t*)act, (const half_t*)dy, P, n_rows, (const half_t*)weights,           \
               (half_t*)dx, grad_w, inv_loss_scale, AttrSrc{nullptr, nullptr, nullptr, 1, 0, -1}, no_epi, dstat);    \
    done = true;                                                                                                     \
  }
  X(1, 1) X(1, 2) X(1, 3) X(2, 1) X(2, 2) X(2, 3) X(8, 1)  // <8, 1> (the density network): the fused path's default since round 5 (-0.17 ms per step; L4D_MLP_RECOMP_SIGMA=0 stores the activations); <6, 2> and wider / deeper spill registers
#undef X
  if (!done && !act) {
    l4d_set_error(1, "l4d_mlp_bwd: act == null (recompute the activations) is only built for in_pad <= 32 and for the 128 -> 64 -> 16 shape");
    return 1;
  }
#define X(IT, NHH)                                                                                                   \
  if (!done && in_pad % 16 == 0 && in_tiles == IT && n_hidden == NHH) {                                              \
    L4D_LAUNCH((mlp_bwd_kernel<IT, NHH, 0, IT, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,         \
                       (const half_t*)x, (const half_t*)act, (const half_t*)dy, P, n_rows, (const half_t*)weights,   \
                       (half_t*)dx, grad_w, inv_loss_scale, AttrSrc{nullptr, nullptr, nullptr, 1, 0, -1}, no_epi, dstat); \
    done = true;                                                                                                     \
  }
  FOR_EACH_CFG(X)
#undef X
#define X(IT, NHH)                                                                                                   \
  if (!done && in_pad % 16 == 0 && in_tiles == IT && n_hidden == NHH) {                                              \
    L4D_LAUNCH((mlp_bwd_kernel<IT, NHH, 0, 6, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,          \
                       (const half_t*)x, (const half_t*)act, (const half_t*)dy, P, n_rows, (const half_t*)weights,   \
                       (half_t*)dx, grad_w, inv_loss_scale, AttrSrc{nullptr, nullptr, nullptr, 1, 0, -1}, no_epi, dstat);   \
    L4D_LAUNCH((mlp_bwd_kernel<IT, NHH, 6, IT, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream,        \
                       (const half_t*)x, (const half_t*)act, (const half_t*)dy, P, n_rows, (const half_t*)weights,   \
                       (half_t*)dx, grad_w, inv_loss_scale, AttrSrc{nullptr, nullptr, nullptr, 1, 0, -1}, no_epi, dstat);   \
    done = true;                                                                                                     \
  }
  FOR_EACH_WIDE_CFG(X)
#undef X
  if (!done) {
    l4d_set_error(1, "l4d_mlp_bwd: unsupported (in_pad, n_hidden)");
    return 1;
  }
  L4D_LAUNCH_CHECK("l4d_mlp_bwd");
  return 0;
}

// ---- attribute networks on a work list, input rows assembled in the kernel (AttrSrc) ------------------------------------
static int attr_src(const int32_t* idx, int32_t T, const void* dir_enc, int32_t n_enc, const void* h, int32_t n_geo, int32_t in_pad,
                    AttrSrc& s, const char* where) {
  if (in_pad != 96 || n_enc % 8 || n_enc / 16 != 4 || n_enc + 16 > in_pad || n_geo != 15 || T <= 0) {
    l4d_set_error(1, where);
    return 1;
  }
  s.idx = idx; s.denc = (const half_t*)dir_enc; s.h = (const half_t*)h; s.T = T; s.n_enc = n_enc; s.cperm = n_enc;
  return 0;
}

extern "C" int l4d_attr_mlp_fwd(const int32_t* idx, const int32_t* count, int64_t cap, int32_t T, const void* dir_enc, int32_t n_enc,
                                const void* h, int32_t n_geo, int32_t in_pad, int32_t n_hidden, const void* weights, void* y,
                                void* act, void* x_rows_out, float* attr_dense, float* attr_compact, int32_t channel, void* stream) {
  if (cap == 0) return 0;
  if (cap >= (int64_t)1 << 31) {
    l4d_set_error(1, "l4d_attr_mlp_fwd: more than 2^31 - 1 samples");
    return 1;
  }
  AttrSrc src;
  if (attr_src(idx, T, dir_enc, n_enc, h, n_geo, in_pad, src, "l4d_attr_mlp_fwd: needs in_pad 96, 64 <= n_enc <= 80 (multiple of 8), n_geo = 15")) return 1;
  const int grid = grid_for((cap + 15) / 16);
  if ((attr_dense == nullptr) != (attr_compact == nullptr) || (!attr_dense && !y) || channel < 0 || channel > 1) {
    l4d_set_error(1, "l4d_attr_mlp_fwd: pass y, or both attr_dense and attr_compact (sigmoid epilogue) with channel 0 / 1");
    return 1;
  }
  const MlpEpi epi{attr_dense, attr_compact, channel};
#define LAUNCH_F(NHH, WX, EP)                                                                                          \
  L4D_LAUNCH((mlp_fwd_kernel<6, NHH, true, WX, EP>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)nullptr, cap, \
             count, (const half_t*)weights, (half_t*)y, (half_t*)act, src, (half_t*)x_rows_out, epi)
#define X(NHH)                                                                                                         \
  if (n_hidden == NHH) {                                                                                               \
    if (x_rows_out && attr_dense) LAUNCH_F(NHH, true, 2);                                                              \
    else if (x_rows_out) LAUNCH_F(NHH, true, 0);                                                                       \
    else if (attr_dense) LAUNCH_F(NHH, false, 2);                                                                      \
    else LAUNCH_F(NHH, false, 0);                                                                                      \
  }
  X(1) X(2) X(3)
#undef X
#undef LAUNCH_F
  if (n_hidden < 1 || n_hidden > 3) { l4d_set_error(1, "l4d_attr_mlp_fwd: n_hidden in 1..3"); return 1; }
  L4D_LAUNCH_CHECK("l4d_attr_mlp_fwd");
  return 0;
}

extern "C" int l4d_attr_mlp_bwd(const void* x_rows, const int32_t* count, int64_t cap, int32_t n_enc, int32_t n_geo, int32_t in_pad,
                                int32_t n_hidden, const void* act, const void* dy, const void* weights, void* dx_tail,
                                float* grad_w, float inv_loss_scale, void* stream) {
  if (cap == 0) return 0;
  AttrSrc src;
  if (attr_src(nullptr, 1, nullptr, n_enc, nullptr, n_geo, in_pad, src, "l4d_attr_mlp_bwd: needs in_pad 96, 64 <= n_enc <= 80 (multiple of 8), n_geo = 15")) return 1;
  int grid = grid_for((cap + 31) / 32);
  if (grid > bwd_grid_cap(in_pad, n_hidden)) grid = bwd_grid_cap(in_pad, n_hidden);
  // the rows l4d_attr_mlp_fwd stored (physical column order: src.cperm permutes the weight columns)
#define X(NHH)                                                                                                          \
  if (n_hidden == NHH)                                                                                                  \
    L4D_LAUNCH((mlp_bwd_kernel<6, NHH, 0, 6, true, false, false, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream,    \
               (const half_t*)x_rows, (const half_t*)act, (const half_t*)dy, cap, count, (const half_t*)weights,        \
               (half_t*)dx_tail, grad_w, inv_loss_scale, src);
  X(1) X(2) X(3)
#undef X
  if (n_hidden < 1 || n_hidden > 3) { l4d_set_error(1, "l4d_attr_mlp_bwd: n_hidden in 1..3"); return 1; }
  L4D_LAUNCH_CHECK("l4d_attr_mlp_bwd");
  return 0;
}

// l4d_attr_mlp_bwd with the input rows assembled here as well (work list + direction encoding + sigma-network rows, as in
// l4d_attr_mlp_fwd) instead of read from a stored [cap, in_pad] copy: the forward then stores no rows at all.
extern "C" int l4d_attr_mlp_bwd_gathered(const int32_t* idx, const int32_t* count, int64_t cap, int32_t T, const void* dir_enc,
                                         int32_t n_enc, const void* h, int32_t n_geo, int32_t in_pad, int32_t n_hidden,
                                         const void* act, const void* dy, const void* weights, void* dx_tail, float* grad_w,
                                         float inv_loss_scale, const float* d_attr, const float* attr_compact, int32_t channel,
                                         float loss_scale, void* dh, int32_t dh_accumulate, void* stream) {
  if (cap == 0) return 0;
  if (cap >= (int64_t)1 << 31) {
    l4d_set_error(1, "l4d_attr_mlp_bwd_gathered: more than 2^31 - 1 samples");
    return 1;
  }
  AttrSrc src;
  if (attr_src(idx, T, dir_enc, n_enc, h, n_geo, in_pad, src, "l4d_attr_mlp_bwd_gathered: needs in_pad 96, 64 <= n_enc <= 80 (multiple of 8), n_geo = 15")) return 1;
  if (n_hidden < 1 || n_hidden > 2) {  // three hidden layers: no register left for the gather (use the stored-rows entry point)
    l4d_set_error(1, "l4d_attr_mlp_bwd_gathered: n_hidden in 1..2");
    return 1;
  }
  const bool epi = d_attr != nullptr;
  if (epi ? (!attr_compact || !dh || channel < 0 || channel > 1) : (!dy || !dx_tail)) {
    l4d_set_error(1, "l4d_attr_mlp_bwd_gathered: pass (dy, dx_tail), or (d_attr, attr_compact, channel 0 / 1, loss_scale, dh)");
    return 1;
  }
  int grid = grid_for((cap + 31) / 32);
  if (grid > bwd_grid_cap(in_pad, n_hidden)) grid = bwd_grid_cap(in_pad, n_hidden);
  const AttrBwdEpi ae{d_attr, attr_compact, (half_t*)dh, channel, dh_accumulate, loss_scale};
  // act == null: the hidden activations are recomputed from the assembled rows (the forward then stores nothing but its
  // sigmoid outputs: 256 B per row and network less to write there and to read here, for 20 more MFMAs per 16 rows)
  if (!act && !epi) {
    l4d_set_error(1, "l4d_attr_mlp_bwd_gathered: act == null (recompute) is only built for the (d_attr, attr_compact, dh) form");
    return 1;
  }
#define X(NHH)                                                                                                               \
  if (n_hidden == NHH) {                                                                                                     \
    if (epi && !act)                                                                                                         \
      L4D_LAUNCH((mlp_bwd_kernel<6, NHH, 0, 6, true, true, true, 4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,   \
                 (const half_t*)nullptr, (const half_t*)nullptr, (const half_t*)nullptr, cap, count, (const half_t*)weights, \
                 (half_t*)nullptr, grad_w, inv_loss_scale, src, ae);                                                         \
    else if (epi)                                                                                                            \
      L4D_LAUNCH((mlp_bwd_kernel<6, NHH, 0, 6, true, false, true, 4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream,  \
                 (const half_t*)nullptr, (const half_t*)act, (const half_t*)nullptr, cap, count, (const half_t*)weights,     \
                 (half_t*)nullptr, grad_w, inv_loss_scale, src, ae);                                                         \
    else                                                                                                                     \
      L4D_LAUNCH((mlp_bwd_kernel<6, NHH, 0, 6, true, false, true, 4>), dim3(grid), dim3(256), 0, (hipStream_t)stream,        \
                 (const half_t*)nullptr, (const half_t*)act, (const half_t*)dy, cap, count, (const half_t*)weights,          \
                 (half_t*)dx_tail, grad_w, inv_loss_scale, src, ae);                                                         \
  }
  X(1) X(2)
#undef X
  L4D_LAUNCH_CHECK("l4d_attr_mlp_bwd_gathered");
  return 0;
}
