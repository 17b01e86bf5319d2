// Device functions of the hex-plane bilinear sampler (ATen grid_sampler_2d semantics: bilinear,
// align_corners=True, border padding), shared by planes.hip and fused.hip.
#pragma once
#include "common.h"

#define MAX_SCALES 8
#define NPLANES 6

struct PlaneDesc {
  int n_scales;
  int res[MAX_SCALES][4];             // x, y, z, t resolution per scale
  int64_t off[MAX_SCALES][NPLANES];   // element offset of plane (scale, comb) in the channel-last arena
};

// comb order of itertools.combinations(range(4), 2): (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)
__device__ __constant__ int COMB_A[NPLANES] = {0, 0, 0, 1, 1, 2};  // -> W axis (grid_sample x)
__device__ __constant__ int COMB_B[NPLANES] = {1, 2, 3, 2, 3, 3};  // -> H axis (grid_sample y)

struct Tap {
  int x0, y0, x1, y1;
  float wx0, wx1, wy0, wy1;
  float mx, my;  // d(ix)/d(coord): (size-1) (= (size-1)/2 * 2), zero where clipped (ATen border rule)
};

__device__ __forceinline__ void axis_tap(float c, int size, int& i0, int& i1, float& w0, float& w1, float& mult) {
  // planes_field.py:76 `coords * 2.0 - 1`, then ATen unnormalize (align_corners) and clip
  float g = c * 2.0f - 1.0f;
  float p = ((g + 1.0f) / 2.0f) * (float)(size - 1);
  const float hi = (float)(size - 1);
  mult = (p > 0.0f && p < hi) ? hi : 0.0f;
  p = fminf(hi, fmaxf(p, 0.0f));
  float f = floorf(p);
  i0 = (int)f;
  i1 = min(i0 + 1, size - 1);  // out-of-range neighbour carries weight 0
  w1 = p - f;
  w0 = (f + 1.0f) - p;
}

// Texel addresses are 32-bit byte offsets from the (wave-uniform) plane base: the loads then use the
// scalar-base + 32-bit-offset form, which costs one VGPR per tap instead of a 64-bit address pair and
// no 64-bit integer VALU work (a plane is at most a few MB).
template <int C>
__device__ __forceinline__ void sample_plane(const float* __restrict__ base, int W, const Tap& t, float out[C]) {
  const float nw = t.wx0 * t.wy0, ne = t.wx1 * t.wy0, sw = t.wx0 * t.wy1, se = t.wx1 * t.wy1;
  const char* b = reinterpret_cast<const char*>(base);
  const uint32_t texel = C * 4u, r0 = (uint32_t)t.y0 * (uint32_t)W, r1 = (uint32_t)t.y1 * (uint32_t)W;
  const float4_t* p00 = reinterpret_cast<const float4_t*>(b + (r0 + (uint32_t)t.x0) * texel);
  const float4_t* p01 = reinterpret_cast<const float4_t*>(b + (r0 + (uint32_t)t.x1) * texel);
  const float4_t* p10 = reinterpret_cast<const float4_t*>(b + (r1 + (uint32_t)t.x0) * texel);
  const float4_t* p11 = reinterpret_cast<const float4_t*>(b + (r1 + (uint32_t)t.x1) * texel);
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    float4_t a = p00[q], b4 = p01[q], c = p10[q], d = p11[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[q * 4 + j] = ((a[j] * nw + b4[j] * ne) + c[j] * sw) + d[j] * se;
  }
}

template <int C>
__device__ __forceinline__ void scatter_plane(float* __restrict__ gbase, const float* __restrict__ base, int W,
                                              const Tap& t, const float gv[C], float& gix, float& giy,
                                              bool want_coord, float pscale = 1.0f) {
  const float nw = t.wx0 * t.wy0, ne = t.wx1 * t.wy0, sw = t.wx0 * t.wy1, se = t.wx1 * t.wy1;
  const size_t o00 = ((size_t)t.y0 * W + t.x0) * C, o01 = ((size_t)t.y0 * W + t.x1) * C;
  const size_t o10 = ((size_t)t.y1 * W + t.x0) * C, o11 = ((size_t)t.y1 * W + t.x1) * C;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const float g = gv[k] * pscale;  // parameter gradients may live in a different (unscaled) domain than gv
    if (g == 0.0f) continue;
    atomicAdd(gbase + o00 + k, g * nw);
    atomicAdd(gbase + o01 + k, g * ne);
    atomicAdd(gbase + o10 + k, g * sw);
    atomicAdd(gbase + o11 + k, g * se);
  }
  if (want_coord) {
    // ATen grid_sampler_2d_backward.  A neighbour index clamped to the border only happens when the
    // coordinate sits exactly on the last texel, where its weight is 0 and the clip mask (mx/my) is 0,
    // so reading the clamped texel instead of ATen's "out of bounds -> 0" changes nothing.
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const float g = gv[k];
      const float v00 = base[o00 + k], v01 = base[o01 + k], v10 = base[o10 + k], v11 = base[o11 + k];
      gix -= v00 * t.wy0 * g;
      giy -= v00 * t.wx0 * g;
      gix += v01 * t.wy0 * g;
      giy -= v01 * t.wx1 * g;
      gix -= v10 * t.wy1 * g;
      giy += v10 * t.wx0 * g;
      gix += v11 * t.wy1 * g;
      giy += v11 * t.wx1 * g;
    }
  }
}


// Bilinear sample that also returns d(sample . gv)/d(ix), d/d(iy) pieces: loads the four taps ONCE (8 x 16 B) and
// gives the interpolated value; call coord_grad_from_taps afterwards with the same tap registers.
template <int C>
struct TapVals {
  float v00[C], v01[C], v10[C], v11[C];
};
template <int C>
__device__ __forceinline__ void load_taps(const float* __restrict__ base, int W, const Tap& t, TapVals<C>& tv, float out[C]) {
  const float nw = t.wx0 * t.wy0, ne = t.wx1 * t.wy0, sw = t.wx0 * t.wy1, se = t.wx1 * t.wy1;
  const float4_t* p00 = reinterpret_cast<const float4_t*>(base + ((size_t)t.y0 * W + t.x0) * C);
  const float4_t* p01 = reinterpret_cast<const float4_t*>(base + ((size_t)t.y0 * W + t.x1) * C);
  const float4_t* p10 = reinterpret_cast<const float4_t*>(base + ((size_t)t.y1 * W + t.x0) * C);
  const float4_t* p11 = reinterpret_cast<const float4_t*>(base + ((size_t)t.y1 * W + t.x1) * C);
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    const float4_t a = p00[q], b = p01[q], c = p10[q], d = p11[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tv.v00[q * 4 + j] = a[j]; tv.v01[q * 4 + j] = b[j]; tv.v10[q * 4 + j] = c[j]; tv.v11[q * 4 + j] = d[j];
      out[q * 4 + j] = ((a[j] * nw + b[j] * ne) + c[j] * sw) + d[j] * se;
    }
  }
}
template <int C>
__device__ __forceinline__ void coord_grad_from_taps(const TapVals<C>& tv, const Tap& t, const float gv[C], float& gix, float& giy) {
  // ATen grid_sampler_2d_backward: gix = sum_k g_k [(v01 - v00) wy0 + (v11 - v10) wy1], giy likewise with x and y swapped
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const float g = gv[k];
    gix += ((tv.v01[k] - tv.v00[k]) * t.wy0 + (tv.v11[k] - tv.v10[k]) * t.wy1) * g;
    giy += ((tv.v10[k] - tv.v00[k]) * t.wx0 + (tv.v11[k] - tv.v01[k]) * t.wx1) * g;
  }
}
