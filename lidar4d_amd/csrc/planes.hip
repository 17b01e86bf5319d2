// Hex-plane (K-Planes) feature sampling for gfx950, replacing the 24 F.grid_sample launches of
// model/planes_field.py:87-141 (bilinear, align_corners=True, border padding, product over the three
// static / three time planes of a scale, concat over scales) with one kernel, forward and backward
// (wrt planes AND wrt sample coordinates: the flow-warped neighbour-frame lookups of
// model/lidar4d.py:158-173 need d/dx).
//
// Layout: the [1,C,H,W] parameters are mirrored channel-last ([H,W,C], C=8 fp32 = 32 B per texel) so a
// bilinear tap is two 16-byte loads instead of 8 strided dwords; l4d_planes_relayout converts both ways
// (parameters -> compute copy, compute-layout gradients -> parameter-layout gradients).
#include "common.h"

#include "planes_dev.h"
#include <algorithm>

template <int C>
__global__ void __launch_bounds__(256) planes_fwd_kernel(PlaneDesc desc, const float* __restrict__ arena,
                                                        const float* __restrict__ xt, int64_t P, int which,
                                                        float* __restrict__ out_s, float* __restrict__ out_d) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float4_t c4 = *reinterpret_cast<const float4_t*>(xt + p * 4);
  const float coord[4] = {c4[0], c4[1], c4[2], c4[3]};
  const int n_out = desc.n_scales * C;
  for (int s = 0; s < desc.n_scales; ++s) {
    float fs[C], fd[C];
    bool first_s = true, first_d = true;
#pragma unroll
    for (int ci = 0; ci < NPLANES; ++ci) {
      const int a = COMB_A[ci], b = COMB_B[ci];
      const bool is_t = (b == 3);
      if ((which == 1 && is_t) || (which == 2 && !is_t)) continue;
      Tap t;
      const int W = desc.res[s][a], H = desc.res[s][b];
      axis_tap(coord[a], W, t.x0, t.x1, t.wx0, t.wx1, t.mx);
      axis_tap(coord[b], H, t.y0, t.y1, t.wy0, t.wy1, t.my);
      float v[C];
      sample_plane<C>(arena + desc.off[s][ci], W, t, v);
      if (is_t) {
#pragma unroll
        for (int k = 0; k < C; ++k) fd[k] = first_d ? v[k] : fd[k] * v[k];
        first_d = false;
      } else {
#pragma unroll
        for (int k = 0; k < C; ++k) fs[k] = first_s ? v[k] : fs[k] * v[k];
        first_s = false;
      }
    }
    if (which != 2) {
      float4_t* o = reinterpret_cast<float4_t*>(out_s + p * n_out + s * C);
#pragma unroll
      for (int q = 0; q < C / 4; ++q) o[q] = float4_t{fs[q * 4], fs[q * 4 + 1], fs[q * 4 + 2], fs[q * 4 + 3]};
    }
    if (which != 1) {
      float4_t* o = reinterpret_cast<float4_t*>(out_d + p * n_out + s * C);
#pragma unroll
      for (int q = 0; q < C / 4; ++q) o[q] = float4_t{fd[q * 4], fd[q * 4 + 1], fd[q * 4 + 2], fd[q * 4 + 3]};
    }
  }
}

template <int C>
__global__ void __launch_bounds__(256) planes_bwd_kernel(PlaneDesc desc, const float* __restrict__ arena,
                                                        const float* __restrict__ xt, int64_t P, int which,
                                                        const float* __restrict__ dout_s,
                                                        const float* __restrict__ dout_d, float* __restrict__ garena,
                                                        float* __restrict__ dxt) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float4_t c4 = *reinterpret_cast<const float4_t*>(xt + p * 4);
  const float coord[4] = {c4[0], c4[1], c4[2], c4[3]};
  const int n_out = desc.n_scales * C;
  float gcoord[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_coord = dxt != nullptr;
  for (int s = 0; s < desc.n_scales; ++s) {
    // recompute the per-plane interpolations of this scale
    Tap taps[NPLANES];
    float v[NPLANES][C];
#pragma unroll
    for (int ci = 0; ci < NPLANES; ++ci) {
      const int a = COMB_A[ci], b = COMB_B[ci];
      const bool is_t = (b == 3);
      if ((which == 1 && is_t) || (which == 2 && !is_t)) continue;
      const int W = desc.res[s][a], H = desc.res[s][b];
      axis_tap(coord[a], W, taps[ci].x0, taps[ci].x1, taps[ci].wx0, taps[ci].wx1, taps[ci].mx);
      axis_tap(coord[b], H, taps[ci].y0, taps[ci].y1, taps[ci].wy0, taps[ci].wy1, taps[ci].my);
      sample_plane<C>(arena + desc.off[s][ci], W, taps[ci], v[ci]);
    }
    float gs[C], gd[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      gs[k] = (which != 2 && dout_s) ? dout_s[p * n_out + s * C + k] : 0.0f;
      gd[k] = (which != 1 && dout_d) ? dout_d[p * n_out + s * C + k] : 0.0f;
    }
#pragma unroll
    for (int ci = 0; ci < NPLANES; ++ci) {
      const int a = COMB_A[ci], b = COMB_B[ci];
      const bool is_t = (b == 3);
      if ((which == 1 && is_t) || (which == 2 && !is_t)) continue;
      // product rule: d(prod)/d(v_ci) = product of the other two planes of the same group
      float gv[C];
      bool any = false;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        float other = 1.0f;
#pragma unroll
        for (int cj = 0; cj < NPLANES; ++cj) {
          if (cj == ci || (COMB_B[cj] == 3) != is_t) continue;
          other *= v[cj][k];
        }
        gv[k] = (is_t ? gd[k] : gs[k]) * other;
        any |= gv[k] != 0.0f;
      }
      if (!any) continue;
      float gix = 0.0f, giy = 0.0f;
      scatter_plane<C>(garena + desc.off[s][ci], arena + desc.off[s][ci], desc.res[s][a], taps[ci], gv, gix, giy,
                       want_coord);
      gcoord[a] += gix * taps[ci].mx;
      gcoord[b] += giy * taps[ci].my;
    }
  }
  if (want_coord) *reinterpret_cast<float4_t*>(dxt + p * 4) = float4_t{gcoord[0], gcoord[1], gcoord[2], gcoord[3]};
}

// [C,H,W] <-> [H,W,C], all planes of all scales in one launch (blockIdx.y = plane); mode 0: channel-last -> [C,H,W],
// 1: [C,H,W] -> channel-last, 2: channel-last ADDED onto [C,H,W] (gradients straight into the parameters' .grad views)
struct RelayoutPlanes {
  float* nchw[MAX_SCALES * NPLANES];
  int64_t off[MAX_SCALES * NPLANES];
  int H[MAX_SCALES * NPLANES], W[MAX_SCALES * NPLANES];
};
__global__ void __launch_bounds__(256) relayout_kernel(RelayoutPlanes pl, float* __restrict__ cl_base, int C, int mode) {
  const int k = blockIdx.y;
  const int H = pl.H[k], W = pl.W[k];
  const int64_t n = (int64_t)C * H * W;
  float* nchw = pl.nchw[k];
  float* cl = cl_base + pl.off[k];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (mode == 1) {  // i indexes cl [H,W,C]
      const int c = i % C;
      const int64_t hw = i / C;
      cl[i] = nchw[(int64_t)c * H * W + hw];
    } else {  // i indexes nchw [C,H,W]
      const int64_t hw = i % ((int64_t)H * W);
      const int c = i / ((int64_t)H * W);
      const float v = cl[hw * C + c];
      nchw[i] = mode == 2 ? nchw[i] + v : v;
    }
  }
}

static int fill_desc(PlaneDesc& d, const int64_t* plane_off, const int32_t* res, int n_scales) {
  if (n_scales > MAX_SCALES) {
    l4d_set_error(1, "planes: too many scales");
    return 1;
  }
  d.n_scales = n_scales;
  for (int s = 0; s < n_scales; ++s) {
    for (int k = 0; k < 4; ++k) d.res[s][k] = res[s * 4 + k];
    for (int c = 0; c < NPLANES; ++c) d.off[s][c] = plane_off[s * NPLANES + c];
  }
  return 0;
}

extern "C" int l4d_planes_relayout(const float* const* planes, const int32_t* res, int32_t n_scales, int32_t C,
                                   float* planes_cl, const int64_t* plane_off, int32_t to_channel_last,
                                   void* stream) {
  if (n_scales > MAX_SCALES || to_channel_last < 0 || to_channel_last > 2) {
    l4d_set_error(1, "l4d_planes_relayout: too many scales / unknown mode");
    return 1;
  }
  RelayoutPlanes pl;
  int64_t max_n = 0;
  for (int s = 0; s < n_scales; ++s)
    for (int c = 0; c < NPLANES; ++c) {
      const int k = s * NPLANES + c;
      static const int CA[NPLANES] = {0, 0, 0, 1, 1, 2}, CB[NPLANES] = {1, 2, 3, 2, 3, 3};  // comb order of the six planes
      pl.W[k] = res[s * 4 + CA[c]];
      pl.H[k] = res[s * 4 + CB[c]];
      pl.off[k] = plane_off[k];
      pl.nchw[k] = const_cast<float*>(planes[k]);
      max_n = std::max<int64_t>(max_n, (int64_t)C * pl.H[k] * pl.W[k]);
    }
  if (max_n == 0) return 0;
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div64(max_n, 256), 512);
  L4D_LAUNCH(relayout_kernel, dim3(gx, n_scales * NPLANES), dim3(256), 0, (hipStream_t)stream, pl, planes_cl, C, to_channel_last);
  L4D_LAUNCH_CHECK("l4d_planes_relayout");
  return 0;
}

extern "C" int l4d_planes_fwd(const float* planes_cl, const int64_t* plane_off, const int32_t* res, int32_t n_scales,
                              int32_t C, const float* xt, int64_t P, int32_t which, float* out_s, float* out_d,
                              void* stream) {
  if (P == 0) return 0;
  if (C != 8) {
    l4d_set_error(1, "planes: C must be 8");
    return 1;
  }
  PlaneDesc d;
  if (fill_desc(d, plane_off, res, n_scales)) return 1;
  L4D_LAUNCH((planes_fwd_kernel<8>), dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream, d,
                     planes_cl, xt, P, which, out_s, out_d);
  L4D_LAUNCH_CHECK("l4d_planes_fwd");
  return 0;
}

extern "C" int l4d_planes_bwd(const float* planes_cl, const int64_t* plane_off, const int32_t* res, int32_t n_scales,
                              int32_t C, const float* xt, int64_t P, int32_t which, const float* dout_s,
                              const float* dout_d, float* grad_cl, float* dxt, void* stream) {
  if (P == 0) return 0;
  if (C != 8) {
    l4d_set_error(1, "planes: C must be 8");
    return 1;
  }
  PlaneDesc d;
  if (fill_desc(d, plane_off, res, n_scales)) return 1;
  L4D_LAUNCH((planes_bwd_kernel<8>), dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream, d,
                     planes_cl, xt, P, which, dout_s, dout_d, grad_cl, dxt);
  L4D_LAUNCH_CHECK("l4d_planes_bwd");
  return 0;
}
