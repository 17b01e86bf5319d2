// Micro-benchmark: LDS atomic flavours (design input).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void lds_kernel(float* out, int per_thread, uint32_t mask) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i <= (int)mask; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = hash32(tid);
  for (int k = 0; k < per_thread; ++k) {
    h = h * 1664525u + 1013904223u;  // cheap LCG so the loop is not ALU-bound
    uint32_t idx = (h >> 8) & mask;
    if (MODE == 0) atomicAdd(&lds[idx], 1.0f);
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int*>(lds) + idx, 1u);
    if (MODE == 2) { float o = lds[idx]; lds[idx] = o + 1.0f; }  // racy RMW, rate reference only
    if (MODE == 3) lds[idx] = 1.0f;                               // plain scattered store
    if (MODE == 4) { half2_t one = {(_Float16)1.f, (_Float16)1.f};
                     __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*)(lds + idx), one); }
    if (MODE == 5) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (idx >> 1), 1ull);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[mask];
}

int main() {
  float* out; hipMalloc(&out, 4096 * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const char* names[] = {"ds_add_f32", "ds_add_u32", "racy ds_read+ds_write", "ds_write_b32 scattered", "ds_pk_add_f16", "ds_add_u64"};
  for (int lds_kb : {32, 128}) {
    uint32_t mask = lds_kb * 256 - 1;
    int blocks = 2048, threads = 512, per_thread = 512;
    double ops = (double)blocks * threads * per_thread;
    for (int mode = 0; mode < 6; ++mode) {
      auto launch = [&] {
        switch (mode) {
          case 0: lds_kernel<0><<<blocks, threads, lds_kb * 1024>>>(out, per_thread, mask); break;
          case 1: lds_kernel<1><<<blocks, threads, lds_kb * 1024>>>(out, per_thread, mask); break;
          case 2: lds_kernel<2><<<blocks, threads, lds_kb * 1024>>>(out, per_thread, mask); break;
          case 3: lds_kernel<3><<<blocks, threads, lds_kb * 1024>>>(out, per_thread, mask); break;
          case 4: lds_kernel<4><<<blocks, threads, lds_kb * 1024>>>(out, per_thread, mask); break;
          case 5: lds_kernel<5><<<blocks, threads, lds_kb * 1024>>>(out, per_thread, mask); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(s); launch(); hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      printf("LDS %3d KB  %-24s %8.3f ms  %9.1f G lane-ops/s  (%.2f lanes/clk/CU @2.1GHz)\n", lds_kb, names[mode], ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
    }
  }
  return 0;
}
