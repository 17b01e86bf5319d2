// Counter calibration (GPU box): kernels that read a KNOWN number of bytes in the access shapes the library's kernels use, each
// launched once under `rocprofv3 --pmc ...` so that FETCH_SIZE / TCC_EA0_RDREQ can be read against the truth:
//   stream16   16 B per lane, consecutive lanes consecutive (wide coalesced stream: the guide's FETCH_SIZE = 1/2 case)
//   stream12   12-byte records, consecutive lanes consecutive records (the sorted scatter's record streams)
//   stream4    4 B per lane (SoA coordinate arrays)
//   rows256    one 16-byte piece per lane out of consecutive 256-byte rows (fp16 [P, 128] matrices read piecewise)
//   rows256x4  the four 16-byte pieces of one 64-byte quarter of such rows, as four loads
//   gather8    8 B per lane at random from a 1 GiB table (every line an HBM miss)
// The buffer (1 GiB) is four times the Infinity Cache; every kernel reads 256 MiB of useful bytes (gather8: 2^24 gathers).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/calib.hip -o tools/ubench/calib_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) calib_stream16(const uint4* __restrict__ src, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_stream12(const uint32_t* __restrict__ src, size_t n_rec, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (size_t)gridDim.x * blockDim.x) { acc ^= src[3 * i] ^ src[3 * i + 1] ^ src[3 * i + 2]; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_stream4(const uint32_t* __restrict__ src, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
  if (acc == 0x12345678u) out[0] = acc;
}
template <int PIECES>
__global__ void __launch_bounds__(256) calib_rows256(const uint4* __restrict__ src, size_t n_rows, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int q = 0; q < PIECES; ++q) { const uint4 v = src[r * 16 + 4 + q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }  // bytes 64 .. 64 + 16 PIECES of the row
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) calib_gather8(const uint2* __restrict__ src, uint32_t mask, size_t n, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint2 v = src[hash32((uint32_t)i) & mask]; acc ^= v.x ^ v.y; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30, useful = (size_t)1 << 28;
  char* buf; uint32_t* out;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0, bytes)); CHECK(hipDeviceSynchronize());
  const int blocks = 256 * 8;
  const size_t n_rows = bytes / 256;  // 2^22 rows of 256 bytes
  // every stream kernel reads its own 256 MiB quarter of the buffer: nothing is cache-warm from the kernel before
  calib_stream16<<<blocks, 256>>>((const uint4*)buf, useful / 16, out);
  calib_stream12<<<blocks, 256>>>((const uint32_t*)(buf + useful), useful / 12, out);
  calib_stream4<<<blocks, 256>>>((const uint32_t*)(buf + 2 * useful), useful / 4, out);
  calib_rows256<1><<<blocks, 256>>>((const uint4*)buf, n_rows, out);
  calib_rows256<4><<<blocks, 256>>>((const uint4*)buf, n_rows, out);
  calib_gather8<<<blocks, 256>>>((const uint2*)buf, (uint32_t)(bytes / 8 - 1), (size_t)1 << 24, out);
  CHECK(hipDeviceSynchronize());
  printf("bytes per launch      useful        as whole 64-B lines\n");
  printf("calib_stream16   %12zu  %12zu\n", useful, useful);
  printf("calib_stream12   %12zu  %12zu\n", useful / 12 * 12, useful);
  printf("calib_stream4    %12zu  %12zu\n", useful, useful);
  printf("calib_rows256<1> %12zu  %12zu   (16 B out of every 256-B row)\n", n_rows * 16, n_rows * 64);
  printf("calib_rows256<4> %12zu  %12zu   (one 64-B line out of every 256-B row, as four 16-B loads)\n", n_rows * 64, n_rows * 64);
  printf("calib_gather8    %12zu  %12zu   (2^24 random 8-B gathers, 1 GiB table)\n", ((size_t)1 << 24) * 8, ((size_t)1 << 24) * 64);
  return 0;
}
