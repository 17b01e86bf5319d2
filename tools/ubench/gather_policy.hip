// Micro-benchmark (round 5): does any cache policy or access width lift the ~290 G lane-loads/s
// ceiling of random table gathers on MI355X?  Random 8-byte gathers (the hash-grid entry) from an L2-resident (2 MB),
// an L2-sized (4 MB) and an Infinity-Cache-resident (32 MB) table:
//   * global_load (flat address) vs buffer_load (descriptor + 32-bit offset);
//   * aux / cache-policy bits of the buffer load: sc0 (1), nt (2), sc1 (16) and their combinations
//     (sc1 / nt loads are L2-served, they bypass the CU's vector L1: /opt/skills/guides/MI355X_MICROARCH.md);
//   * 4, 8 and 16 bytes per lane; one 16-byte load that covers an aligned PAIR of entries; two independent 8-byte loads.
// Rate = lane-loads per second chip-wide; "lines/s" counts one 128-byte line per lane-load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }


// MODE 0: global_load, W bytes;  MODE 1: buffer_load with AUX, W bytes
template <int MODE, int W, int AUX>
__global__ void __launch_bounds__(256) gather_kernel(const uint32_t* __restrict__ table, uint32_t mask, uint32_t bytes, int iters, uint32_t* __restrict__ out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(table), 0, (int)bytes, 0x00020000);
  uint32_t acc = 0;
  for (int k = 0; k < iters; ++k) {
    uint32_t idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) idx[u] = hash32(tid * 977u + (k * 8 + u) * 0x9e3779b9u) & mask;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        if (W == 4) { acc ^= table[idx[u]]; }
        if (W == 8) { const uint2 v = *reinterpret_cast<const uint2*>(table + (size_t)idx[u] * 2); acc ^= v.x ^ v.y; }
        if (W == 16) { const uint4 v = *reinterpret_cast<const uint4*>(table + (size_t)idx[u] * 4); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
      } else {
        if (W == 4) { acc ^= (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(idx[u] * 4u), 0, AUX); }
        if (W == 8) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(idx[u] * 8u), 0, AUX); acc ^= v.x ^ v.y; }
        if (W == 16) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(idx[u] * 16u), 0, AUX); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int W, int AUX>
static int run(const char* name, uint32_t* table, uint32_t* out, size_t table_bytes) {
  const uint32_t mask = (uint32_t)(table_bytes / W) - 1;
  const int blocks = 256 * 16, iters = 32;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  gather_kernel<MODE, W, AUX><<<blocks, 256>>>(table, mask, (uint32_t)table_bytes, 4, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  gather_kernel<MODE, W, AUX><<<blocks, 256>>>(table, mask, (uint32_t)table_bytes, iters, out);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double lanes = (double)blocks * 256 * iters * 8;
  printf("%-40s table %5.1f MB: %7.1f G lane-loads/s  %6.1f clk/wave-instr/CU  %7.1f GB/s useful  %6.2f TB/s of 128-B lines\n", name,
         table_bytes / 1048576.0, lanes / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / (lanes / 64), lanes * W / ms / 1e6, lanes * 128 / ms / 1e9);
  return 0;
}

int main() {
  uint32_t *table, *out;
  const size_t max_bytes = 64u << 20;
  CHECK(hipMalloc(&table, max_bytes)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(table, 1, max_bytes));
  for (size_t bytes : {(size_t)2 << 20, (size_t)4 << 20, (size_t)32 << 20}) {
    if (run<0, 8, 0>("global_load 8 B", table, out, bytes)) return 1;
    if (run<0, 4, 0>("global_load 4 B", table, out, bytes)) return 1;
    if (run<0, 16, 0>("global_load 16 B", table, out, bytes)) return 1;
    if (run<1, 8, 0>("buffer_load 8 B", table, out, bytes)) return 1;
    if (run<1, 8, 1>("buffer_load 8 B sc0", table, out, bytes)) return 1;
    if (run<1, 8, 2>("buffer_load 8 B nt", table, out, bytes)) return 1;
    if (run<1, 8, 3>("buffer_load 8 B sc0 nt", table, out, bytes)) return 1;
    if (run<1, 8, 16>("buffer_load 8 B sc1", table, out, bytes)) return 1;
    if (run<1, 8, 17>("buffer_load 8 B sc0 sc1", table, out, bytes)) return 1;
    if (run<1, 8, 18>("buffer_load 8 B sc1 nt", table, out, bytes)) return 1;
    if (run<1, 8, 19>("buffer_load 8 B sc0 sc1 nt", table, out, bytes)) return 1;
    if (run<1, 4, 0>("buffer_load 4 B", table, out, bytes)) return 1;
    if (run<1, 4, 16>("buffer_load 4 B sc1", table, out, bytes)) return 1;
    if (run<1, 16, 0>("buffer_load 16 B (aligned entry pair)", table, out, bytes)) return 1;
    if (run<1, 16, 16>("buffer_load 16 B sc1", table, out, bytes)) return 1;
    if (run<1, 16, 2>("buffer_load 16 B nt", table, out, bytes)) return 1;
  }
  return 0;
}
