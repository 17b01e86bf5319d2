// Range image <-> point cloud conversions on the device, replacing the numpy round trips of
// utils/convert.py:4-156 that model/runner.py:764-767, model/simulator.py:137-142 and utils/metrics.py:253-254 make
// once per rendered frame (GPU -> host -> python loop -> GPU).
//
//   pano_to_lidar : [H,W] range (+ intensity) -> the non-empty pixels as [N,4] points, in row-major pixel order
//                   (np.where order), convert.py:99-137.  Ordered compaction without atomics: per-workgroup counts,
//                   then every workgroup sums the counts in front of it and ranks its own pixels with a ballot scan.
//   lidar_to_pano : [N,4] points -> [H,W] nearest range + its intensity, convert.py:4-66.  The reference walks the
//                   points one by one (python loop, "closest wins, first of equals stays"); here one 64-bit
//                   atomicMin per point on (range bits << 32 | point index) gives the same winner in any order.
// Arithmetic is fp32 like numpy's on float32 inputs; index rounding is round-half-to-even like python's round().
#include "common.h"

#define CV_THREADS 1024
#define CV_PI 3.14159265358979323846

// direction of pixel (row j, column i): convert.py:112-123 == data/base_dataset.py:82-93
__device__ __forceinline__ void pixel_dir(int j, int i, int H, int W, float fov_up, float fov, float& dx, float& dy, float& dz) {
  const float pi = (float)CV_PI;
  const float beta = -((float)i - (float)W / 2.0f) / (float)W * 2.0f * pi;
  const float alpha = (fov_up - (float)j / (float)H * fov) / 180.0f * pi;
  const float ca = cosf(alpha);
  dx = ca * cosf(beta);
  dy = ca * sinf(beta);
  dz = sinf(alpha);
}

__global__ void __launch_bounds__(CV_THREADS) pano_count_kernel(const float* __restrict__ pano, int64_t n, int32_t* __restrict__ counts) {
  __shared__ int wave_cnt[CV_THREADS / 64];
  const int64_t i = (int64_t)blockIdx.x * CV_THREADS + threadIdx.x;
  const bool keep = i < n && pano[i] != 0.0f;
  const unsigned long long b = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < CV_THREADS / 64; ++w) s += wave_cnt[w];
    counts[blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(CV_THREADS) pano_emit_kernel(const float* __restrict__ pano, const float* __restrict__ inten,
                                                              int H, int W, float fov_up, float fov,
                                                              const int32_t* __restrict__ counts, float* __restrict__ out,
                                                              int32_t* __restrict__ total) {
  __shared__ int wave_cnt[CV_THREADS / 64];
  __shared__ int part[CV_THREADS / 64];
  __shared__ int base_s;
  const int64_t n = (int64_t)H * W;
  // points emitted by the workgroups in front of this one
  int acc = 0;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += CV_THREADS) acc += counts[k];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  const int64_t i = (int64_t)blockIdx.x * CV_THREADS + threadIdx.x;
  const float r = i < n ? pano[i] : 0.0f;
  const bool keep = r != 0.0f;
  const unsigned long long b = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wave] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < CV_THREADS / 64; ++w) s += part[w];
    base_s = s;
    if (blockIdx.x == gridDim.x - 1) {
      int mine = 0;
      for (int w = 0; w < CV_THREADS / 64; ++w) mine += wave_cnt[w];
      *total = s + mine;
    }
  }
  __syncthreads();
  if (!keep) return;
  int pos = base_s + __popcll(b & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
  const int j = (int)(i / W), c = (int)(i - (int64_t)j * W);
  float dx, dy, dz;
  pixel_dir(j, c, H, W, fov_up, fov, dx, dy, dz);
  float4 p;
  p.x = dx * r;
  p.y = dy * r;
  p.z = dz * r;
  p.w = inten ? inten[i] : 0.0f;
  reinterpret_cast<float4*>(out)[pos] = p;
}

extern "C" int64_t l4d_pano_to_lidar_workspace(int32_t H, int32_t W) { return (ceil_div64((int64_t)H * W, CV_THREADS) + 1) * 4; }

extern "C" int l4d_pano_to_lidar(const float* pano, const float* intensities, int32_t H, int32_t W, double fov_up, double fov,
                                 float* points, int32_t* count, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t n = (int64_t)H * W;
  if (n == 0) { l4d_fill_async(count, 0u, 4, stream); return 0; }
  const unsigned blocks = (unsigned)ceil_div64(n, CV_THREADS);
  int32_t* counts = (int32_t*)workspace;
  L4D_LAUNCH(pano_count_kernel, dim3(blocks), dim3(CV_THREADS), 0, stream, pano, n, counts);
  L4D_LAUNCH(pano_emit_kernel, dim3(blocks), dim3(CV_THREADS), 0, stream, pano, intensities, (int)H, (int)W, (float)fov_up,
                     (float)fov, (const int32_t*)counts, points, count);
  L4D_LAUNCH_CHECK("l4d_pano_to_lidar");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// off_a = fov_down/180*pi, step_c = 2 pi / W, step_r = fov/180*pi/H: python-float (fp64) expressions in the reference,
// rounded to fp32 when they meet the float32 point data (convert.py:50-53)
__global__ void lidar_bin_kernel(const float* __restrict__ pts, int64_t n, int H, int W, float off_a, float step_c, float step_r,
                                 float max_depth, unsigned long long* __restrict__ best) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = reinterpret_cast<const float4*>(pts)[i];
  // np.linalg.norm on float32: sqrt of the fp32 sum of squares (convert.py:33)
  const float dist = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
  if (!(dist < max_depth)) return;  // convert.py:44
  const float beta = (float)CV_PI - atan2f(p.y, p.x);
  const float alpha = atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)) + off_a;
  const float cf = rintf(beta / step_c);
  const float rf = rintf((float)H - alpha / step_r);
  if (!(rf >= 0.0f && rf < (float)H && cf >= 0.0f && cf < (float)W)) return;  // convert.py:54
  if (dist == 0.0f) return;  // a zero range is the "unset" marker of the reference's image
  const int r = (int)rf, c = (int)cf;
  const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned int)i;
  atomicMin(best + (size_t)r * W + c, key);
}

__global__ void lidar_unpack_kernel(const unsigned long long* __restrict__ best, const float* __restrict__ pts, int64_t npix,
                                    float* __restrict__ pano, float* __restrict__ inten) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const unsigned long long k = best[i];
  if (k == ~0ull) {
    pano[i] = 0.0f;
    if (inten) inten[i] = 0.0f;
    return;
  }
  pano[i] = __uint_as_float((unsigned int)(k >> 32));
  if (inten) inten[i] = pts[(size_t)(unsigned int)(k & 0xffffffffu) * 4 + 3];
}

__global__ void pano_fill_kernel(unsigned long long* __restrict__ best, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) best[i] = ~0ull;
}

extern "C" int64_t l4d_lidar_to_pano_workspace(int32_t H, int32_t W) { return (int64_t)H * W * 8; }

extern "C" int l4d_lidar_to_pano(const float* points, int64_t n, int32_t H, int32_t W, double fov_up, double fov, float max_depth,
                                 float* pano, float* intensities, void* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t npix = (int64_t)H * W;
  if (npix == 0) return 0;
  if (n >= (1ll << 32)) { l4d_set_error(1, "l4d_lidar_to_pano: more than 2^32 points"); return 1; }
  unsigned long long* best = (unsigned long long*)workspace;
  // filled by a kernel: a 0xff hipMemsetAsync node made a captured graph fault at its second replay (see chamfer.hip)
  L4D_LAUNCH(pano_fill_kernel, dim3((unsigned)ceil_div64(npix, 256)), dim3(256), 0, stream, best, npix);
  if (n > 0)
    L4D_LAUNCH(lidar_bin_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, stream, points, n, (int)H, (int)W,
                       (float)((fov - fov_up) / 180 * CV_PI), (float)(2 * CV_PI / W), (float)(fov / 180 * CV_PI / H), max_depth, best);
  L4D_LAUNCH(lidar_unpack_kernel, dim3((unsigned)ceil_div64(npix, 256)), dim3(256), 0, stream,
                     (const unsigned long long*)best, points, npix, pano, intensities);
  L4D_LAUNCH_CHECK("l4d_lidar_to_pano");
  return 0;
}
