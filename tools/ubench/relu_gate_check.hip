// relu_gate (csrc/mlp_dev.h) against the compare form on random packed halfs, on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../lidar4d_amd/csrc/mlp_dev.h"
__global__ void k(const uint4* v, const uint4* h, uint4* out, uint4* ref, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  h8 vv = __builtin_bit_cast(h8, v[i]), hh = __builtin_bit_cast(h8, h[i]);
  f4 lo, hi;
  for (int e = 0; e < 4; ++e) { lo[e] = (float)vv[e]; hi[e] = (float)vv[4 + e]; }
  h8 r = relu_gate(lo, hi, hh);
  out[i] = __builtin_bit_cast(uint4, r);
  h8 q;
  for (int e = 0; e < 8; ++e) q[e] = hh[e] > (_Float16)0.0f ? (_Float16)(float)vv[e] : (_Float16)0.0f;
  ref[i] = __builtin_bit_cast(uint4, q);
}
int main() {
  const int n = 1 << 16;
  uint4 *v, *h, *o, *r;
  hipMalloc(&v, n * 16); hipMalloc(&h, n * 16); hipMalloc(&o, n * 16); hipMalloc(&r, n * 16);
  uint4* hv = (uint4*)malloc(n * 16); uint4* hh = (uint4*)malloc(n * 16);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned w[8];
    for (int j = 0; j < 4; ++j) {  // finite halfs of either sign (an infinity now and then)
      unsigned a = rand() & 0xFFFF, b = rand() & 0xFFFF;
      if ((a & 0x7C00) == 0x7C00) a &= 0xFC00;
      if ((b & 0x7C00) == 0x7C00) b &= 0xFC00;
      w[j] = a | (b << 16);
    }
    for (int j = 0; j < 4; ++j) {  // activations: non-negative halfs, a third of them zero, some denormal
      unsigned a = rand() % 3 == 0 ? 0u : (rand() % 7 == 0 ? (unsigned)(rand() & 0x3FF) : (unsigned)(rand() & 0x7BFF));
      unsigned b = rand() % 3 == 0 ? 0u : (unsigned)(rand() & 0x7BFF);
      w[4 + j] = a | (b << 16);
    }
    hv[i] = make_uint4(w[0], w[1], w[2], w[3]); hh[i] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  hipMemcpy(v, hv, n * 16, hipMemcpyHostToDevice); hipMemcpy(h, hh, n * 16, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(v, h, o, r, n);
  uint4* ho = (uint4*)malloc(n * 16); uint4* hr = (uint4*)malloc(n * 16);
  hipMemcpy(ho, o, n * 16, hipMemcpyDeviceToHost); hipMemcpy(hr, r, n * 16, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned* a = (const unsigned*)&ho[i]; const unsigned* b = (const unsigned*)&hr[i];
    for (int j = 0; j < 4; ++j) if (a[j] != b[j]) { if (bad < 5) printf("mismatch at %d.%d: gate %08x ref %08x v %08x h %08x\n", i, j, a[j], b[j], ((unsigned*)&hv[i])[j], ((unsigned*)&hh[i])[j]); ++bad; }
  }
  printf("relu_gate: %ld mismatching dwords of %d\n", bad, n * 4);
  return bad != 0;
}
