// The small per-step stages either side of the render path, ONE launch each.
//
// A training step of the reference (model/runner.py:166-253) draws a ray batch (data/base_dataset.py:15-102), renders it, and
// evaluates its losses with a few dozen element-wise torch operations on [N]-sized tensors: on MI355X that was ~250 launches of
// ~5 us per step (1.3 ms of kernel time + the gaps between them, profiles/r03_kernel_stats_final.txt) -- a third of the
// reference's own 1,024-ray step.  A captured graph does not shrink them (the step is GPU-bound); fusing does.
//
//   l4d_lidar_ray_batch     drawn pixels -> rays_o / rays_d / ground-truth pixels            (base_dataset.py:72-102 + the gather
//                                                                                              of kitti360_dataset.py:181-187)
//   l4d_lidar_losses        the three primary losses, their gradients, and the two point sets of the ray-chamfer term
//                           (runner.py:179-219)
//   l4d_ray_chamfer_grad    mean of the chamfer distances + their gradient wrt the rendered depth (runner.py:215-220 under autograd)
//   l4d_scale_buffers       g *= s[0] for the saved gradients (s = the upstream gradient, i.e. the loss scale, on the device)
//
// All arithmetic is written in the reference's operation order (this file is compiled without fma contraction), so that the
// results agree with the torch path to the last bit wherever torch itself is deterministic (everything but its reductions).
#include <algorithm>

#include "common.h"

// ---- ray batch -------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lidar_ray_batch_kernel(const int64_t* __restrict__ rows, const int64_t* __restrict__ cols, int n,
                                                              const float* __restrict__ pose, float fov_up, float fov, int H, int W,
                                                              const float* __restrict__ image, float* __restrict__ rays_o,
                                                              float* __restrict__ rays_d, float* __restrict__ gt,
                                                              int64_t* __restrict__ inds) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int64_t r = rows[k], c = cols[k] % W;
  const int64_t ind = r * W + c;
  inds[k] = ind;
  // base_dataset.py:82-93: i = column, j = row (floats); beta = -(i - W/2) / W * 2 * pi; alpha = (fov_up - j / H * fov) / 180 * pi
  const float i = (float)c, j = (float)r;
  const float pi = 3.14159265358979323846f;
  const float beta = -(i - (float)((double)W / 2.0)) / (float)W * 2.0f * pi;
  const float alpha = (fov_up - j / (float)H * fov) / 180.0f * pi;
  const float ca = cosf(alpha), sa = sinf(alpha), cb = cosf(beta), sb = sinf(beta);
  const float d[3] = {ca * cb, ca * sb, sa};
#pragma unroll
  for (int a = 0; a < 3; ++a) {  // rays_d = directions @ R^T, rays_o = translation
    rays_d[k * 3 + a] = d[0] * pose[a * 4 + 0] + d[1] * pose[a * 4 + 1] + d[2] * pose[a * 4 + 2];
    rays_o[k * 3 + a] = pose[a * 4 + 3];
  }
  if (image) {
#pragma unroll
    for (int a = 0; a < 3; ++a) gt[k * 3 + a] = image[ind * 3 + a];
  }
}

extern "C" int l4d_lidar_ray_batch(const int64_t* rows, const int64_t* cols, int32_t n, const float* pose, float fov_up, float fov,
                                   int32_t H, int32_t W, const float* image, float* rays_o, float* rays_d, float* gt, int64_t* inds,
                                   void* stream) {
  if (n <= 0) return 0;
  if (H <= 0 || W <= 0) { l4d_set_error(1, "l4d_lidar_ray_batch: empty image"); return 1; }
  L4D_LAUNCH(lidar_ray_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, cols, n, pose, fov_up, fov, H, W,
             image, rays_o, rays_d, gt, inds);
  L4D_LAUNCH_CHECK("l4d_lidar_ray_batch");
  return 0;
}

// ---- primary losses --------------------------------------------------------------------------------------------------------
// fixed-order sum of one value per thread over the workgroup (1024 threads): the same result every run
__device__ __forceinline__ float block_sum_1024(float v, float* red /* [1024] */) {
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// fixed-order sum of one value per thread over a 256-thread workgroup
__device__ __forceinline__ float block_sum_256(float v, float* red /* [256] */) {
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// loss[0] = (accumulate ? loss[0] : 0) + coef * (partial[0] + partial[1] + ...), summed in index order by one workgroup: the
// per-block partial sums of the kernels below become ONE number that is the same every run (no floating-point atomics)
__global__ void __launch_bounds__(256) sum_partials_kernel(const float* __restrict__ partial, int n, float coef, int accumulate,
                                                           float* __restrict__ loss) {
  __shared__ float red[256];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const float total = block_sum_256(acc, red);
  if (threadIdx.x == 0) loss[0] = (accumulate ? loss[0] : 0.0f) + coef * total;
}

// runner.py:179-213 with the default criteria (L1 depth, MSE ray-drop on the label-smoothed mask, MSE intensity), all masked by
// the ground-truth ray-drop and summed (partial[block] = the block's share); pts: [2][n][3] predicted / ground-truth points along
// the rays in metres (runner.py:215-218)
__global__ void __launch_bounds__(256) lidar_losses_kernel(const float* __restrict__ depth, const float* __restrict__ image,
                                                          const float* __restrict__ gt, const float* __restrict__ rays_d, int n,
                                                          float alpha_d, float alpha_r, float alpha_i, float smooth, float scale,
                                                          float* __restrict__ partial, float* __restrict__ g_depth,
                                                          float* __restrict__ g_image, float* __restrict__ pts) {
  __shared__ float red[256];
  float acc = 0.0f;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) {
    const float m = gt[k * 3 + 0];
    const float gt_i = gt[k * 3 + 1] * m, gt_d = gt[k * 3 + 2] * m;
    const float p_r = image[k * 2 + 0], p_i = image[k * 2 + 1] * m, p_d = depth[k] * m;
    const float gs = fminf(fmaxf(m, smooth), 1.0f - smooth);
    const float ed = p_d - gt_d, er = p_r - gs, ei = p_i - gt_i;
    acc = alpha_d * fabsf(ed) + alpha_r * (er * er) + alpha_i * (ei * ei);
    g_depth[k] = alpha_d * (ed > 0.0f ? 1.0f : ed < 0.0f ? -1.0f : 0.0f) * m;
    g_image[k * 2 + 0] = alpha_r * (2.0f * er);
    g_image[k * 2 + 1] = alpha_i * (2.0f * ei) * m;
    if (pts) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float rd = rays_d[k * 3 + a];
        pts[k * 3 + a] = rd * p_d / scale;
        pts[(n + k) * 3 + a] = rd * gt_d / scale;
      }
    }
  }
  const float total = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// partial: scratch of l4d_glue_workspace(n) floats
extern "C" int64_t l4d_glue_workspace(int32_t n) { return (int64_t)(n > 0 ? (n + 255) / 256 : 1); }

extern "C" int l4d_lidar_losses(const float* depth, const float* image, const float* gt, const float* rays_d, int32_t n, float alpha_d,
                                float alpha_r, float alpha_i, float smooth, float scale, float* loss, float* g_depth, float* g_image,
                                float* pts, float* partial, void* stream) {
  if (n < 0) { l4d_set_error(1, "l4d_lidar_losses: negative ray count"); return 1; }
  const int blocks = (n + 255) / 256;
  if (blocks > 0)
    L4D_LAUNCH(lidar_losses_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, depth, image, gt, rays_d, n, alpha_d, alpha_r, alpha_i,
               smooth, scale, partial, g_depth, g_image, pts);
  L4D_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, blocks, 1.0f, 0, loss);
  L4D_LAUNCH_CHECK("l4d_lidar_losses");
  return 0;
}

// loss[0] += coef * sum(dist1 + dist2); g_depth += coef * d(sum)/d(depth) through p_k = rays_d_k * (depth_k * m_k) / scale:
//   dist1_k = |p_k - q_idx1[k]|^2  -> 2 (p_k - q_a) . rays_d_k * m_k / scale
//   dist2_j = |q_j - p_idx2[j]|^2  -> 2 (p_b - q_j) . rays_d_b * m_b / scale into b   (scattered: atomics, 2 n of them)
__global__ void __launch_bounds__(256) ray_chamfer_grad_kernel(const float* __restrict__ pts, const float* __restrict__ rays_d,
                                                              const float* __restrict__ gt, const float* __restrict__ dist1,
                                                              const float* __restrict__ dist2, const int32_t* __restrict__ idx1,
                                                              const int32_t* __restrict__ idx2, int n, float coef, float scale,
                                                              float* __restrict__ partial, float* __restrict__ g_depth) {
  __shared__ float red[256];
  const float* p = pts;
  const float* q = pts + (size_t)n * 3;
  float acc = 0.0f;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) {
    acc = dist1[k] + dist2[k];
    const int a = idx1[k];
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dot += (p[k * 3 + c] - q[a * 3 + c]) * rays_d[k * 3 + c];
    atomicAdd(g_depth + k, coef * 2.0f * dot * gt[k * 3] / scale);
    const int b = idx2[k];
    dot = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dot += (p[b * 3 + c] - q[k * 3 + c]) * rays_d[b * 3 + c];
    atomicAdd(g_depth + b, coef * 2.0f * dot * gt[b * 3] / scale);
  }
  const float total = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

extern "C" int l4d_ray_chamfer_grad(const float* pts, const float* rays_d, const float* gt, const float* dist1, const float* dist2,
                                    const int32_t* idx1, const int32_t* idx2, int32_t n, float coef, float scale, float* loss,
                                    float* g_depth, float* partial, void* stream) {
  if (n <= 0) return 0;
  const int blocks = (n + 255) / 256;
  L4D_LAUNCH(ray_chamfer_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pts, rays_d, gt, dist1, dist2, idx1, idx2, n, coef,
             scale, partial, g_depth);
  L4D_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, blocks, coef, 1, loss);
  L4D_LAUNCH_CHECK("l4d_ray_chamfer_grad");
  return 0;
}

// out_a[i] = a[i] * s[0], out_b[i] = b[i] * s[0] (either pair may be empty)
__global__ void __launch_bounds__(256) scale_buffers_kernel(const float* __restrict__ a, float* __restrict__ out_a, int64_t na,
                                                            const float* __restrict__ b, float* __restrict__ out_b, int64_t nb,
                                                            const float* __restrict__ s) {
  const float f = s[0];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) out_a[i] = a[i] * f;
  else if (i < na + nb) out_b[i - na] = b[i - na] * f;
}

extern "C" int l4d_scale_buffers(const float* a, float* out_a, int64_t na, const float* b, float* out_b, int64_t nb, const float* s,
                                 void* stream) {
  if (na + nb <= 0) return 0;
  L4D_LAUNCH(scale_buffers_kernel, dim3((unsigned)ceil_div64(na + nb, 256)), dim3(256), 0, (hipStream_t)stream, a, out_a, na, b, out_b,
             nb, s);
  L4D_LAUNCH_CHECK("l4d_scale_buffers");
  return 0;
}

// ---- scene-flow consistency loss (runner.py:222-253) ---------------------------------------------------------------------------
// xt[k] = [(pc[k] + bound) / (2 bound), t] (lidar4d.py:133-137: flow() normalises the points and appends the call's time)
__global__ void __launch_bounds__(256) flow_xt_kernel(const float* __restrict__ pc, int n, const float* __restrict__ t, float bound,
                                                      float* __restrict__ xt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float tt = t[0], den = 2.0f * bound;
  float4_t o;
  o[0] = (pc[k * 3 + 0] + bound) / den;
  o[1] = (pc[k * 3 + 1] + bound) / den;
  o[2] = (pc[k * 3 + 2] + bound) / den;
  o[3] = tt;
  *reinterpret_cast<float4_t*>(xt + (size_t)k * 4) = o;
}

extern "C" int l4d_flow_xt(const float* pc, int32_t n, const float* t, float bound, float* xt, void* stream) {
  if (n <= 0) return 0;
  L4D_LAUNCH(flow_xt_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pc, n, t, bound, xt);
  L4D_LAUNCH_CHECK("l4d_flow_xt");
  return 0;
}

// out[v][k] = pc[k] + float(y16[k][col0[v] .. col0[v] + 2]) * step[v]: the point cloud warped by the forward / backward flow over
// one or two frame steps (runner.py:233-247), up to four variants per launch
struct FlowWarps {
  int n_variants;
  int col0[4];
  float step[4];
};
__global__ void __launch_bounds__(256) flow_warp_kernel(const float* __restrict__ pc, const half_t* __restrict__ y16, int n, FlowWarps w,
                                                        float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float p[3] = {pc[k * 3 + 0], pc[k * 3 + 1], pc[k * 3 + 2]};
  const uint4 raw = *reinterpret_cast<const uint4*>(y16 + (size_t)k * 16);
  const half_t* h = reinterpret_cast<const half_t*>(&raw);
  for (int v = 0; v < w.n_variants; ++v) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((size_t)v * n + k) * 3 + c] = p[c] + h2f(h[w.col0[v] + c]) * w.step[v];
  }
}

extern "C" int l4d_flow_warp(const float* pc, const void* y16, int32_t n, int32_t n_variants, const int32_t* col0, const float* step,
                             float* out, void* stream) {
  if (n <= 0 || n_variants <= 0) return 0;
  if (n_variants > 4) { l4d_set_error(1, "l4d_flow_warp: at most 4 variants"); return 1; }
  FlowWarps w;
  w.n_variants = n_variants;
  for (int v = 0; v < 4; ++v) {
    w.col0[v] = v < n_variants ? col0[v] : 0;
    w.step[v] = v < n_variants ? step[v] : 0.0f;
    if (w.col0[v] < 0 || w.col0[v] > 5) { l4d_set_error(1, "l4d_flow_warp: column out of range"); return 1; }
  }
  L4D_LAUNCH(flow_warp_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pc, (const half_t*)y16, n, w, out);
  L4D_LAUNCH_CHECK("l4d_flow_warp");
  return 0;
}

// One chamfer term 0.5 (sum dist1 + sum dist2) between the warped cloud p [n,3] and a neighbour frame's cloud q [m,3]
// (runner.py:236-247): partial[block] = this block's share of sum(dist1) + sum(dist2) (summed in block order by
// l4d_flow_loss_finish: the loss value is the same every run), and its gradient wrt the flow OUTPUT columns col0 .. col0 + 2 of the
// points: d/dp_i = (p_i - q_idx1[i]) from dist1 and (p_b - q_j) into b = idx2[j] from dist2, times step (p = pc + flow * step).
__global__ void __launch_bounds__(256) flow_chamfer_grad_kernel(const float* __restrict__ p, int n, const float* __restrict__ q, int m,
                                                                const float* __restrict__ dist1, const float* __restrict__ dist2,
                                                                const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
                                                                float step, int col0, float* __restrict__ dy,
                                                                float* __restrict__ partial) {
  __shared__ float red[256];
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.0f;
  if (k < n) {
    acc += dist1[k];
    const int a = idx1[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(dy + (size_t)k * 6 + col0 + c, step * (p[k * 3 + c] - q[a * 3 + c]));
  }
  if (k < m) {
    acc += dist2[k];
    const int b = idx2[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(dy + (size_t)b * 6 + col0 + c, step * (p[b * 3 + c] - q[k * 3 + c]));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
#pragma unroll
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

extern "C" int l4d_flow_chamfer_grad(const float* p, int32_t n, const float* q, int32_t m, const float* dist1, const float* dist2,
                                     const int32_t* idx1, const int32_t* idx2, float step, int32_t col0, float* dy, float* partial,
                                     void* stream) {
  if (n <= 0 || m <= 0) return 0;
  if (col0 != 0 && col0 != 3) { l4d_set_error(1, "l4d_flow_chamfer_grad: col0 must be 0 (forward flow) or 3 (backward flow)"); return 1; }
  const int blocks = (std::max(n, m) + 255) / 256;
  L4D_LAUNCH(flow_chamfer_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, q, m, dist1, dist2, idx1, idx2, step, col0,
             dy, partial);
  L4D_LAUNCH_CHECK("l4d_flow_chamfer_grad");
  return 0;
}

// loss[0] = 0.5 * sum(partial[0 .. n_partial)) + w_ground * sum |y_g[:, 0..5]|; dy_g [ng,6] = w_ground * sign(y_g) (runner.py:249-252:
// 0.001 * L1 of the flow of the ground points); amax[0] / amax[1] = max |dy| / max |dy_g| (what the backward's fp16 range needs)
__global__ void __launch_bounds__(1024) flow_loss_finish_kernel(const float* __restrict__ partial, int n_partial, const half_t* __restrict__ yg,
                                                               int ng, float w_ground, float* __restrict__ dy_g, const float* __restrict__ dy,
                                                               int64_t n_dy, float* __restrict__ loss, float* __restrict__ amax) {
  __shared__ float red[1024];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n_partial; i += 1024) acc += partial[i];
  const float cham = block_sum_1024(acc, red);
  acc = 0.0f;
  for (int k = threadIdx.x; k < ng; k += 1024) {
    const uint4 raw = *reinterpret_cast<const uint4*>(yg + (size_t)k * 16);
    const half_t* h = reinterpret_cast<const half_t*>(&raw);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float v = h2f(h[c]);
      acc += fabsf(v);
      dy_g[(size_t)k * 6 + c] = w_ground * (v > 0.0f ? 1.0f : v < 0.0f ? -1.0f : 0.0f);
    }
  }
  const float l1 = block_sum_1024(acc, red);
  float mx = 0.0f;
  {  // (n_dy = 6 n is even and dy is 8-byte aligned: two values per load)
    const float2_t* dy2 = reinterpret_cast<const float2_t*>(dy);
    for (int64_t i = threadIdx.x; i < n_dy / 2; i += 1024) {
      const float2_t v = dy2[i];
      mx = (nonfinite(v[0]) || nonfinite(v[1])) ? __builtin_inff() : fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1])));
    }
    if ((n_dy & 1) && threadIdx.x == 0) mx = nonfinite(dy[n_dy - 1]) ? __builtin_inff() : fmaxf(mx, fabsf(dy[n_dy - 1]));
  }
  red[threadIdx.x] = mx;
  __syncthreads();
#pragma unroll
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = 0.5f * cham + w_ground * l1;
    amax[0] = red[0];
    amax[1] = ng > 0 ? w_ground : 0.0f;
  }
}

extern "C" int l4d_flow_loss_finish(const float* partial, int32_t n_partial, const void* y_ground16, int32_t ng, float w_ground, float* dy_g,
                                    const float* dy, int64_t n_dy, float* loss, float* amax, void* stream) {
  L4D_LAUNCH(flow_loss_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, n_partial, (const half_t*)y_ground16, ng,
             w_ground, dy_g, dy, n_dy, loss, amax);
  L4D_LAUNCH_CHECK("l4d_flow_loss_finish");
  return 0;
}

// dy16[k][0..5] = fp16(dy[k][0..5] * g * s), columns 6..15 zero, with s = 2^k the power of two that puts amax * |g| * s into
// [2^11, 2^12) (flow_field.py: fp16 adjoints normalised per call ON THE DEVICE; k clamped to +-40); inv_out[0] = 1 / s.  A
// non-finite g or amax gives non-finite dy16: the overflow reaches the parameter gradients and the scaler (common.h f2h_grad).
__global__ void __launch_bounds__(256) flow_dy16_kernel(const float* __restrict__ dy, int n, const float* __restrict__ g,
                                                        const float* __restrict__ amax, half_t* __restrict__ dy16, float* __restrict__ inv_out) {
  const float gg = g[0];
  const float a = fmaxf(amax[0] * fabsf(gg), 1e-30f);
  const float kf = fminf(fmaxf(floorf(log2f(4096.0f / a)), -40.0f), 40.0f);
  const float s = exp2f(kf) * gg;  // (power of two) x upstream gradient
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) inv_out[0] = exp2f(-kf);
  if (k >= n) return;
  half_t o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = c < 6 ? f2h_grad(dy[(size_t)k * 6 + c] * s) : (half_t)0.0f;
  uint4* dst = reinterpret_cast<uint4*>(dy16 + (size_t)k * 16);
  dst[0] = reinterpret_cast<uint4*>(o)[0];
  dst[1] = reinterpret_cast<uint4*>(o)[1];
}

extern "C" int l4d_flow_dy16(const float* dy, int32_t n, const float* g, const float* amax, void* dy16, float* inv_out, void* stream) {
  if (n <= 0) return 0;
  L4D_LAUNCH(flow_dy16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, n, g, amax, (half_t*)dy16, inv_out);
  L4D_LAUNCH_CHECK("l4d_flow_dy16");
  return 0;
}

// y[i] += a[0] * x[i] (a on the device; n a multiple of 4 is not required)
__global__ void __launch_bounds__(256) axpy_dev_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, const float* __restrict__ a) {
  const float f = a[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (v != 0.0f) y[i] += f * v;  // (the private gradient buffers are sparse: most of the arena range is not written at all)
  }
}

extern "C" int l4d_axpy_dev(float* y, const float* x, int64_t n, const float* a, void* stream) {
  if (n <= 0) return 0;
  const int64_t blocks = std::min<int64_t>(8192, ceil_div64(n, 256));
  L4D_LAUNCH(axpy_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, x, n, a);
  L4D_LAUNCH_CHECK("l4d_axpy_dev");
  return 0;
}
