// Micro-benchmark: vector-memory GATHER issue rate on MI355X by access width, coherence and table size (design input for
// the hash-grid / hex-plane lookups): how many cycles a wave-level gather instruction occupies the CU's address path when
// (a) every lane touches its own cache line, (b) runs of R consecutive lanes share an address (consecutive samples of a ray
// in one cell / texel), (c) a second load hits the neighbouring entry of the same line.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// W = bytes per lane (8 or 16); RUN = lanes sharing one address; PAIR = 1: also load the adjacent entry (idx ^ 1)
template <int W, int RUN, int PAIR>
__global__ void __launch_bounds__(256) gather_kernel(const uint32_t* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int k = 0; k < iters; ++k) {
    uint32_t idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) idx[u] = hash32((tid / RUN) * 977u + (k * 8 + u) * 0x9e3779b9u) & mask;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (W == 8) {
        const uint2 v = *reinterpret_cast<const uint2*>(table + (size_t)idx[u] * 2);
        acc ^= v.x ^ v.y;
        if (PAIR) { const uint2 q = *reinterpret_cast<const uint2*>(table + (size_t)(idx[u] ^ 1u) * 2); acc ^= q.x + q.y; }
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(table + (size_t)idx[u] * 4);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
        if (PAIR) { const uint4 q = *reinterpret_cast<const uint4*>(table + (size_t)(idx[u] ^ 1u) * 4); acc ^= q.x + q.y + q.z + q.w; }
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;  // keep the loads alive
}

template <int W, int RUN, int PAIR>
static int run(const char* name, uint32_t* table, uint32_t* out, size_t table_bytes) {
  const uint32_t mask = (uint32_t)(table_bytes / W) - 1;
  const int blocks = 256 * 16, iters = 32;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  gather_kernel<W, RUN, PAIR><<<blocks, 256>>>(table, mask, 4, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  gather_kernel<W, RUN, PAIR><<<blocks, 256>>>(table, mask, iters, out);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double lanes = (double)blocks * 256 * iters * 8 * (PAIR ? 2 : 1);
  const double winstr = lanes / 64;
  // cycles of one CU's address path per wave instruction, at 256 CUs x 2.4 GHz
  printf("%-34s table %6.1f MB: %8.1f G lane-loads/s  %6.1f clk/wave-instr/CU  %7.1f GB/s useful\n", name, table_bytes / 1048576.0,
         lanes / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / winstr, lanes * W / ms / 1e6);
  return 0;
}

int main() {
  uint32_t *table, *out;
  const size_t max_bytes = 64u << 20;
  CHECK(hipMalloc(&table, max_bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(table, 1, max_bytes));
  for (size_t bytes : {(size_t)256 << 10, (size_t)2 << 20, (size_t)32 << 20}) {
    if (run<8, 1, 0>("8 B random", table, out, bytes)) return 1;
    if (run<16, 1, 0>("16 B random", table, out, bytes)) return 1;
    if (run<8, 1, 1>("8 B random + adjacent entry", table, out, bytes)) return 1;
    if (run<16, 1, 1>("16 B random + adjacent (32 B texel)", table, out, bytes)) return 1;
    if (run<8, 4, 0>("8 B runs of 4 lanes", table, out, bytes)) return 1;
    if (run<8, 16, 0>("8 B runs of 16 lanes", table, out, bytes)) return 1;
    if (run<16, 4, 0>("16 B runs of 4 lanes", table, out, bytes)) return 1;
    if (run<16, 16, 0>("16 B runs of 16 lanes", table, out, bytes)) return 1;
    if (run<16, 16, 1>("16 B runs of 16 + adjacent", table, out, bytes)) return 1;
  }
  return 0;
}
