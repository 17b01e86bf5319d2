// Brute-force chamfer distance (nearest neighbour both ways) for gfx950, replacing the reference's only in-tree
// native op: utils/chamfer3D/chamfer3D.cu:11-194 (NmDistanceKernel / NmDistanceGradKernel) behind
// utils/chamfer3D/dist_chamfer_3D.py:31-83.  Same contract: squared distance + index of the nearest point, first
// minimum wins; backward 2 * grad * (p - q) into both clouds.
//
// The reference launches dim3(32,16) x 512 threads and only blockIdx.y strides the points of one cloud, so with its
// batch size 1 just 16 workgroups work.  Here the candidate cloud is split into segments as well
// (grid = query tiles x segments x batch, thousands of workgroups), each workgroup streams its segment through LDS,
// and the per-segment winners are merged with ONE 64-bit atomicMin per (query, segment) on the packed key
// (distance bits << 32 | index): non-negative floats order like their bit patterns and ties fall to the smaller index,
// which is exactly "first minimum wins".
#include <algorithm>

#include "common.h"

#define CH_THREADS 256
#define CH_TILE 1024

__global__ void __launch_bounds__(CH_THREADS) chamfer_nn_kernel(const float* __restrict__ a, int n, const float* __restrict__ b, int m,
                                                               int seg_len, unsigned long long* __restrict__ best) {
  __shared__ float tile[CH_TILE * 3];
  const int batch = blockIdx.z;
  const float* A = a + (size_t)batch * n * 3;
  const float* B = b + (size_t)batch * m * 3;
  const int q = blockIdx.x * CH_THREADS + threadIdx.x;
  const bool valid = q < n;
  float x1 = 0.f, y1 = 0.f, z1 = 0.f;
  if (valid) { x1 = A[q * 3]; y1 = A[q * 3 + 1]; z1 = A[q * 3 + 2]; }
  const int seg0 = blockIdx.y * seg_len, seg1 = min(m, seg0 + seg_len);
  float best_d = 3.4e38f;
  int best_i = seg0;
  for (int t0 = seg0; t0 < seg1; t0 += CH_TILE) {
    const int cnt = min(CH_TILE, seg1 - t0);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt * 3; j += CH_THREADS) tile[j] = B[(size_t)t0 * 3 + j];
    __syncthreads();
    if (valid) {
#pragma unroll 4
      for (int k = 0; k < cnt; ++k) {
        const float dx = tile[k * 3] - x1, dy = tile[k * 3 + 1] - y1, dz = tile[k * 3 + 2] - z1;
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best_d) { best_d = d; best_i = t0 + k; }
      }
    }
  }
  if (valid && seg1 > seg0) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(best_d) << 32) | (unsigned int)best_i;
    atomicMin(best + (size_t)batch * n + q, key);
  }
}

// the workspace is initialised by a kernel, not hipMemsetAsync: a captured step whose graph held this memset node faulted at its
// second replay (ROCm 7.2, linear graph; tools/graph_probe.py), the same graph without the node replays
__global__ void chamfer_fill_kernel(unsigned long long* __restrict__ best, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) best[i] = ~0ull;
}

__global__ void chamfer_unpack_kernel(const unsigned long long* __restrict__ best, int64_t total, float* __restrict__ dist,
                                      int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned long long k = best[i];
  dist[i] = __uint_as_float((unsigned int)(k >> 32));
  idx[i] = (int32_t)(k & 0xffffffffu);
}

// chamfer3D.cu:154-173: g = 2 grad_dist[i]; grad_a[i] += g (a_i - b_j); grad_b[j] -= g (a_i - b_j), j = idx[i]
__global__ void chamfer_grad_kernel(const float* __restrict__ a, int n, const float* __restrict__ b, int m,
                                    const float* __restrict__ grad_dist, const int32_t* __restrict__ idx,
                                    float* __restrict__ grad_a, float* __restrict__ grad_b) {
  const int batch = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t ia = ((size_t)batch * n + i) * 3;
  const int j = idx[(size_t)batch * n + i];
  const size_t ib = ((size_t)batch * m + j) * 3;
  const float g = grad_dist[(size_t)batch * n + i] * 2.0f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float v = g * (a[ia + d] - b[ib + d]);
    atomicAdd(grad_a + ia + d, v);
    atomicAdd(grad_b + ib + d, -v);
  }
}

static int seg_for(int n, int m, int b) {
  // enough workgroups to fill 256 CUs several times over, segments not shorter than one LDS tile
  const int q_tiles = (n + CH_THREADS - 1) / CH_THREADS;
  int segs = std::max(1, 2048 / std::max(1, q_tiles * b));
  segs = std::min(segs, (m + CH_TILE - 1) / CH_TILE);
  return std::max(1, segs);
}

extern "C" int64_t l4d_chamfer_workspace(int32_t b, int32_t n, int32_t m) { return (int64_t)b * ((int64_t)n + m) * 8; }

extern "C" int l4d_chamfer_fwd(const float* xyz1, const float* xyz2, int32_t b, int32_t n, int32_t m, float* dist1, float* dist2,
                               int32_t* idx1, int32_t* idx2, void* workspace, void* stream_) {
  if (b == 0 || (n == 0 && m == 0)) return 0;
  if (n == 0 || m == 0) { l4d_set_error(1, "l4d_chamfer_fwd: both clouds must be non-empty"); return 1; }
  hipStream_t stream = (hipStream_t)stream_;
  unsigned long long* best1 = (unsigned long long*)workspace;
  unsigned long long* best2 = best1 + (size_t)b * n;
  const int64_t total = (int64_t)b * ((int64_t)n + m);
  L4D_LAUNCH(chamfer_fill_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, stream, best1, total);
  {
    const int segs = seg_for(n, m, b), seg_len = (m + segs - 1) / segs;
    L4D_LAUNCH(chamfer_nn_kernel, dim3((n + CH_THREADS - 1) / CH_THREADS, (m + seg_len - 1) / seg_len, b), dim3(CH_THREADS), 0,
                       stream, xyz1, n, xyz2, m, seg_len, best1);
  }
  {
    const int segs = seg_for(m, n, b), seg_len = (n + segs - 1) / segs;
    L4D_LAUNCH(chamfer_nn_kernel, dim3((m + CH_THREADS - 1) / CH_THREADS, (n + seg_len - 1) / seg_len, b), dim3(CH_THREADS), 0,
                       stream, xyz2, m, xyz1, n, seg_len, best2);
  }
  L4D_LAUNCH(chamfer_unpack_kernel, dim3((unsigned)ceil_div64((int64_t)b * n, 256)), dim3(256), 0, stream, best1, (int64_t)b * n,
                     dist1, idx1);
  L4D_LAUNCH(chamfer_unpack_kernel, dim3((unsigned)ceil_div64((int64_t)b * m, 256)), dim3(256), 0, stream, best2, (int64_t)b * m,
                     dist2, idx2);
  L4D_LAUNCH_CHECK("l4d_chamfer_fwd");
  return 0;
}

extern "C" int l4d_chamfer_bwd(const float* xyz1, const float* xyz2, int32_t b, int32_t n, int32_t m, const float* grad_dist1,
                               const float* grad_dist2, const int32_t* idx1, const int32_t* idx2, float* grad_xyz1, float* grad_xyz2,
                               void* stream_) {
  if (b == 0 || n == 0 || m == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  L4D_LAUNCH(chamfer_grad_kernel, dim3((n + 255) / 256, b), dim3(256), 0, stream, xyz1, n, xyz2, m, grad_dist1, idx1, grad_xyz1,
                     grad_xyz2);
  L4D_LAUNCH(chamfer_grad_kernel, dim3((m + 255) / 256, b), dim3(256), 0, stream, xyz2, m, xyz1, n, grad_dist2, idx2, grad_xyz2,
                     grad_xyz1);
  L4D_LAUNCH_CHECK("l4d_chamfer_bwd");
  return 0;
}
