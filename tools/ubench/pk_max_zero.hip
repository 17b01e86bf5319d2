#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  unsigned a = 0x80008000u, z = 0u;  // {-0, -0}, {+0, +0}
  asm volatile("" : "+v"(a), "+v"(z));
  h2 ha = __builtin_bit_cast(h2, a), hz = __builtin_bit_cast(h2, z);
  h2 r1 = __builtin_elementwise_max(ha, hz), r2 = __builtin_elementwise_max(hz, ha);
  o[0] = __builtin_bit_cast(unsigned, r1);
  o[1] = __builtin_bit_cast(unsigned, r2);
  unsigned r3, r4;
  asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r3) : "v"(a), "v"(z));
  asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r4) : "v"(z), "v"(a));
  o[2] = r3; o[3] = r4;
  // tiny negative converted: -1e-9f -> f16
  float t = -1e-9f; asm volatile("" : "+v"(t));
  h2 c = {(_Float16)t, (_Float16)t};
  o[4] = __builtin_bit_cast(unsigned, c);
  h2 m = __builtin_elementwise_max(c, hz);
  o[5] = __builtin_bit_cast(unsigned, m);
}
int main() {
  unsigned* d; hipMalloc(&d, 64); k<<<1, 1>>>(d); unsigned h[8]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) printf("o[%d] = %08x\n", i, h[i]);
  return 0;
}
