// Per-point fused field evaluation of LiDAR4D.density (model/lidar4d.py:139-179) for gfx950:
// hex-planes at (x,t) + time planes at the two flow-warped neighbour-frame points, static 3-D hash grid,
// the three 2-D x time HashGridT stacks at the current and both neighbour frames, the 0.5/0.25/0.25 blends and
// the concat -- written straight into the sigma MLP's fp16 input row [planes_s | planes_d | hash_s | hash_d | 1].
// One thread owns one sample point; no intermediate [P, .] tensor is materialised (the reference creates ~40).
// The backward kernel is the exact adjoint: scatters into plane / hash gradients and returns d(flow).
//
// This is the direct-gather formulation (every table entry is fetched through L1/L2).
#include "field_dev.h"

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
// (profiles/r01_pmc_WRITE_SIZE_c3.txt), so the row is staged in LDS (272-byte row pitch: 16-B aligned, spreads the
// lanes' rows over the banks) and each wave then writes its 64 rows as full 16-B-per-lane coalesced stores.
#define ENC_THREADS 128
#define ENC_PITCH 136  // halfs per staged row (128 + 8 pad)
__global__ void __launch_bounds__(ENC_THREADS) density_encode_fwd_kernel(FieldDesc fd, const float* __restrict__ xt,
                                                                        const half_t* __restrict__ flow16,
                                                                        const float* __restrict__ tinfo, int64_t P,
                                                                        half_t* __restrict__ X, int in_pad) {
  constexpr int C = 8;
  __shared__ __attribute__((aligned(16))) half_t stage[ENC_THREADS * ENC_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wave_p0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;
  const int64_t pr = wave_p0 + lane;
  const int64_t p = pr < P ? pr : P - 1;
  const float4_t c4 = *reinterpret_cast<const float4_t*>(xt + p * 4);
  const float t0 = tinfo[0], t1 = tinfo[1], t2 = tinfo[2];
  const bool has_fwd = tinfo[3] != 0.0f, has_bwd = tinfo[4] != 0.0f;
  float fl[8];
  {
    uint4 u = *reinterpret_cast<const uint4*>(flow16 + p * 16);
    const half_t* h = reinterpret_cast<const half_t*>(&u);
#pragma unroll
    for (int k = 0; k < 8; ++k) fl[k] = h2f(h[k]);
  }
  const float x0[4] = {c4[0], c4[1], c4[2], t0};
  const float x1[4] = {c4[0] + fl[0], c4[1] + fl[1], c4[2] + fl[2], t1};
  const float x2[4] = {c4[0] + fl[3], c4[1] + fl[4], c4[2] + fl[5], t2};
  half_t* row = stage + (wave * 64 + lane) * ENC_PITCH;
  const int nS = fd.planes.n_scales;

  // ---- hex-planes (planes_field.py:87-141; blend lidar4d.py:175) ----
  for (int s = 0; s < nS; ++s) {
    float ps[C], d0[C], d1[C], d2[C];
    planes_group<C>(fd, s, x0, false, ps);
    planes_group<C>(fd, s, x0, true, d0);
    if (has_fwd) planes_group<C>(fd, s, x1, true, d1);
    if (has_bwd) planes_group<C>(fd, s, x2, true, d2);
    float pd[C];
#pragma unroll
    for (int k = 0; k < C; ++k) pd[k] = 0.5f * d0[k] + 0.25f * ((has_fwd ? d1[k] : d0[k]) + (has_bwd ? d2[k] : d0[k]));
    store8h(row + s * C, ps);
    store8h(row + (nS + s) * C, pd);
  }
  int col = 2 * nS * C;

  // ---- static 3-D hash grid (hash_field.py:141-144) ----
  {
    const float xs[3] = {x0[0], x0[1], x0[2]};
    for (int lvl = 0; lvl < fd.hs.n_levels; ++lvl) {
      float a[4];
      level_lookup<3, 4>(fd.hs_table + (size_t)fd.hs.offset[lvl] * 4, fd.hs.scale[lvl], fd.hs.res[lvl], fd.hs.size[lvl],
                         (fd.hs.hashed_mask >> lvl) & 1u, xs, a);
      half4_t h;
#pragma unroll
      for (int f = 0; f < 4; ++f) h[f] = f2h(a[f]);
      *reinterpret_cast<half4_t*>(row + col + lvl * 4) = h;
    }
    col += fd.hs.n_levels * 4;
  }

  // ---- dynamic HashGridT stacks at the current and the two warped neighbour frames (lidar4d.py:145,157-176) ----
  const TimeCoef tc0 = time_coef(t0, fd.n_slices), tc1 = time_coef(t1, fd.n_slices), tc2 = time_coef(t2, fd.n_slices);
#pragma unroll
  for (int plane = 0; plane < 3; ++plane) {
    const int ca = plane == 2 ? 1 : 0, cb = plane == 0 ? 1 : 2;  // xy, xz, yz
    const float q0[2] = {x0[ca], x0[cb]}, q1[2] = {x1[ca], x1[cb]}, q2[2] = {x2[ca], x2[cb]};
    const int L = fd.hd[plane].n_levels;
    for (int lvl = 0; lvl < L; ++lvl) {
      const float r0 = hash_t_level(fd, plane, lvl, tc0, q0);
      const float r1 = has_fwd ? hash_t_level(fd, plane, lvl, tc1, q1) : r0;
      const float r2 = has_bwd ? hash_t_level(fd, plane, lvl, tc2, q2) : r0;
      row[col + lvl] = f2h(0.5f * r0 + 0.25f * (r1 + r2));
    }
    col += L;
  }
  for (; col < in_pad; ++col) row[col] = (half_t)1.0f;  // tcnn pads the network input with ones

  __syncthreads();  // rows of this wave complete (and visible) before the cooperative copy-out
  const int chunks = in_pad / 8;  // 16-byte chunks per row
  const half_t* wstage = stage + wave * 64 * ENC_PITCH;
  for (int idx = lane; idx < 64 * chunks; idx += 64) {
    const int r = idx / chunks, c = idx - r * chunks;
    const int64_t grow = wave_p0 + r;
    if (grow < P) *reinterpret_cast<uint4*>(X + grow * in_pad + c * 8) = *reinterpret_cast<const uint4*>(wstage + r * ENC_PITCH + c * 8);
  }
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
  hipLaunchKernelGGL(time_setup_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, num_frames, tinfo);
  L4D_LAUNCH_CHECK("l4d_time_setup");
  return 0;
}

extern "C" int l4d_sample_rays_xt(const float* rays_o, const float* rays_d, const float* lin, const float* noise,
                                  const float* t, int64_t N, int32_t T, float near, float far, float bound, float* z_vals,
                                  float* xt, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(sample_rays_xt_kernel, dim3((unsigned)ceil_div64(N * T, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, lin, noise, t, N, T, near, far, bound, z_vals, xt);
  L4D_LAUNCH_CHECK("l4d_sample_rays_xt");
  return 0;
}

extern "C" int l4d_density_encode_fwd(const l4d_field_desc* f, const float* xt, const void* flow16, const float* tinfo,
                                      int64_t P, void* X, int32_t in_pad, void* stream) {
  if (P == 0) return 0;
  FieldDesc d;
  if (make_field(f, d)) return 1;
  if (l4d_field_width(f) > in_pad || in_pad % 8 || in_pad > 128) {
    l4d_set_error(1, "l4d_density_encode_fwd: in_pad too small for the field width (or not a multiple of 8)");
    return 1;
  }
  hipLaunchKernelGGL(density_encode_fwd_kernel, dim3((unsigned)ceil_div64(P, ENC_THREADS)), dim3(ENC_THREADS), 0, (hipStream_t)stream, d, xt,
                     (const half_t*)flow16, tinfo, P, (half_t*)X, in_pad);
  L4D_LAUNCH_CHECK("l4d_density_encode_fwd");
  return 0;
}

