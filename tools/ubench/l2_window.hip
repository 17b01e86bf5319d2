// Micro-benchmark: how large may a randomly gathered table be before an XCD's 4 MB L2 stops holding it -- alone, next to the
// streaming traffic a level-major lookup pass would carry (16 B in + 8 B out per 8 gathers per lane), with that traffic marked
// non-temporal, and with one private table per XCD (level l pinned to XCD l % 8).  Design input for DESIGN.md section 7, item 1:
// the forward encode is bound by lines that miss L2 (64 G/s against 268-292 G/s for hits; gather.hip), and the fine hash levels
// are 4 MB tables.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// STREAM: 0 none, 1 plain loads / stores, 2 non-temporal;  PER_XCD: 1 = block b gathers from table number b % 8
template <int STREAM, int PER_XCD>
__global__ void __launch_bounds__(256) window_kernel(const uint2* __restrict__ tables, uint32_t n_entries, size_t table_stride, int iters,
                                                     const uint4* __restrict__ in, uint2* __restrict__ outs, uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint2* table = tables + (PER_XCD ? (size_t)(blockIdx.x & 7) * table_stride : 0);
  uint32_t acc = 0;
  for (int k = 0; k < iters; ++k) {
    const size_t s = (size_t)k * gridDim.x * blockDim.x + tid;
    uint32_t salt = 0;
    if (STREAM == 1) { const uint4 c = in[s]; salt = c.x ^ c.w; }
    if (STREAM == 2) { const u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in) + s); salt = c.x ^ c.w; }
    uint32_t idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) idx[u] = (uint32_t)(((uint64_t)hash32(tid * 977u + (k * 8 + u) * 0x9e3779b9u + salt) * n_entries) >> 32);
    uint32_t r = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { const uint2 v = table[idx[u]]; r ^= v.x ^ v.y; }
    acc ^= r;
    if (STREAM == 1) outs[s] = make_uint2(r, acc);
    if (STREAM == 2) { const u32x2 o = {r, acc}; __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(outs) + s); }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int STREAM, int PER_XCD>
static int run(const char* name, uint2* tables, size_t table_bytes, uint4* in, uint2* outs, uint32_t* sink) {
  const uint32_t n_entries = (uint32_t)(table_bytes / 8);
  const int blocks = 256 * 8, iters = 48;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  window_kernel<STREAM, PER_XCD><<<blocks, 256>>>(tables, n_entries, (64u << 20) / 8, 6, in, outs, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  window_kernel<STREAM, PER_XCD><<<blocks, 256>>>(tables, n_entries, (64u << 20) / 8, iters, in, outs, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double lanes = (double)blocks * 256 * iters * 8;
  printf("%-44s table %5.1f MB%s: %7.1f G gathers/s   (%5.2f ms; streamed %.0f MB)\n", name, table_bytes / 1048576.0, PER_XCD ? " per XCD" : "        ",
         lanes / ms / 1e6, ms, STREAM ? (double)blocks * 256 * iters * 24 / 1e6 : 0.0);
  return 0;
}

int main() {
  uint2* tables; uint4* in; uint2* outs; uint32_t* sink;
  const size_t n_stream = (size_t)256 * 8 * 256 * 48;
  CHECK(hipMalloc(&tables, (size_t)8 * (64u << 20))); CHECK(hipMalloc(&in, n_stream * 16)); CHECK(hipMalloc(&outs, n_stream * 8)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(tables, 1, (size_t)8 * (64u << 20))); CHECK(hipMemset(in, 2, n_stream * 16));
  const size_t sizes[] = {(size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)3 << 20, (size_t)4 << 20, (size_t)6 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20};
  for (size_t bytes : sizes) {
    if (run<0, 0>("gathers only, one table for all XCDs", tables, bytes, in, outs, sink)) return 1;
    if (run<1, 0>("+ 16 B in / 8 B out per 8 gathers", tables, bytes, in, outs, sink)) return 1;
    if (run<2, 0>("+ the same, non-temporal", tables, bytes, in, outs, sink)) return 1;
    if (run<2, 1>("+ non-temporal stream, a table per XCD", tables, bytes, in, outs, sink)) return 1;
  }
  return 0;
}
