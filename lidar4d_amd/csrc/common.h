// Shared device/host helpers for the gfx950 kernels.  Compiled with -ffp-contract=off: every
// fused multiply-add in these sources is an explicit fmaf(), so coordinate / index arithmetic
// rounds exactly as the oracle's (and the reference's torch) separate mul+add do.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/lidar4d_hip.h"

#define L4D_WAVE 64

// (library-internal helpers: not part of the C ABI, not exported from the shared object)
#define L4D_INTERNAL extern "C" __attribute__((visibility("hidden")))
L4D_INTERNAL void l4d_set_error(int code, const char* where);

#define L4D_LAUNCH_CHECK(where)                         \
  do {                                                  \
    hipError_t e__ = hipGetLastError();                 \
    if (e__ != hipSuccess) {                            \
      l4d_set_error((int)e__, where);                   \
      return (int)e__;                                  \
    }                                                   \
  } while (0)

// Every kernel launch of the library goes through this macro: when profiling is switched on (l4d_profile_enable, used by
// bench.py's per-kernel pass) a pair of HIP events is recorded around the launch ON THE LAUNCH STREAM, so the durations
// bench.py reports are per KERNEL (same names as rocprofv3's kernel trace), also for entry points that launch several.
// side streams (capi.cpp): l4d_side_fork(from, i) -> stream i continuing from `from`; l4d_side_join(into, i)
extern "C" int l4d_streams_mask(void);
extern "C" void* l4d_side_fork(void* from, int32_t i);
extern "C" int l4d_side_join(void* into, int32_t i);
L4D_INTERNAL int l4d_prof_begin(const char* kernel, void* stream);
L4D_INTERNAL void l4d_prof_end(int idx, void* stream);
L4D_INTERNAL void l4d_trace_sync(const char* kernel, void* stream);
#define L4D_LAUNCH(kernel, grid, block, lds, stream, ...)                    \
  do {                                                                       \
    const int prof_idx__ = l4d_prof_begin(#kernel, (void*)(stream));         \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);       \
    if (prof_idx__ >= 0) l4d_prof_end(prof_idx__, (void*)(stream));          \
    else if (prof_idx__ == -2) l4d_trace_sync(#kernel, (void*)(stream));     \
  } while (0)

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));  // arithmetic on it compiles to v_pk_*_f32 (2 fp32 ops per lane-instruction)

__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
__device__ __forceinline__ half_t f2h(float f) { return (half_t)f; }  // round-to-nearest-even
// fp16 narrowing of BACKWARD quantities.  Not saturated on purpose: a gradient that leaves the fp16 range becomes inf
// (and inf * 0 = nan further down), every kernel of the adjoint chain hands non-finite values on to the parameter
// gradients it produces, and the optimiser side (l4d_grad_nonfinite_check, or torch's GradScaler.unscale_ in the
// reference's own loop, runner.py:506-508) then skips the step and lowers the loss scale -- tiny-cuda-nn / autocast
// semantics.  A saturating clamp here would hide the overflow and let a growing loss scale corrupt gradients silently.
__device__ __forceinline__ half_t f2h_grad(float f) { return (half_t)f; }
__device__ __forceinline__ bool nonfinite(float x) { return !(fabsf(x) <= 3.402823466e38f); }  // inf or nan

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware tile order.  The dispatcher deals workgroups round-robin to the 8 XCDs (block b runs on XCD b % 8), each
// with its own 4 MB L2.  Mapping block b to tile (b % 8) * (n / 8) + b / 8 gives every XCD one CONTIGUOUS eighth of the
// tiles -- with rays sorted by pixel neighbourhood that is one sector of the scene, so the table lines a sector
// touches are fetched into one L2 instead of all eight.  Launch xcd_grid(n) blocks; tiles >= n are idle.
static inline int64_t xcd_grid(int64_t n_tiles) { return ceil_div64(n_tiles, 8) * 8; }
#ifdef __HIPCC__
// Device memory is initialised and small device values are moved by KERNELS, never by hipMemsetAsync / hipMemcpyAsync: a captured
// training step whose graph held such a node (the chamfer workspace's 0xff fill) faulted at its second replay on ROCm 7.2 when the
// graph was a single chain of nodes, and replayed once the node was a kernel (tools/graph_probe.py).
static __global__ void l4d_fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
static __global__ void l4d_copy_u32_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// fill `bytes` (a multiple of 4) at p with the 32-bit pattern v
static inline void l4d_fill_async(void* p, uint32_t v, int64_t bytes, hipStream_t stream) {
  const int64_t n = bytes / 4;
  if (n <= 0) return;
  const int64_t blocks = (n + 255) / 256;
  L4D_LAUNCH(l4d_fill_u32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, (uint32_t*)p, v, n);
}
static inline void l4d_copy_words_async(void* dst, const void* src, int n_words, hipStream_t stream) {
  if (n_words <= 0) return;
  L4D_LAUNCH(l4d_copy_u32_kernel, dim3((unsigned)((n_words + 63) / 64)), dim3(64), 0, stream, (uint32_t*)dst, (const uint32_t*)src, n_words);
}
__device__ __forceinline__ int64_t xcd_tile(int64_t block, int64_t n_blocks /* multiple of 8 */) {
  return (block & 7) * (n_blocks >> 3) + (block >> 3);
}
#endif

// Device copy of l4d_grid_desc passed by value as a kernel argument (236 bytes).
struct GridDesc {
  int n_levels;
  uint32_t hashed_mask;
  float scale[L4D_MAX_LEVELS];
  uint32_t res[L4D_MAX_LEVELS];
  uint32_t size[L4D_MAX_LEVELS];
  uint32_t offset[L4D_MAX_LEVELS];
};

static inline GridDesc make_grid_desc(const l4d_grid_desc* d) {
  GridDesc g;
  g.n_levels = d->n_levels;
  g.hashed_mask = d->hashed_mask;
  for (int i = 0; i < L4D_MAX_LEVELS; ++i) {
    g.scale[i] = d->scale[i];
    g.res[i] = d->res[i];
    g.size[i] = d->size[i];
    g.offset[i] = d->offset[i];
  }
  return g;
}

// Lagrange basis of the reference's interpT (model/hash_field.py:65-74) at nodes {0,1/3,2/3,1}.
// Products in the reference's order: for j, prod over m != j (ascending m) of (t - T[m]) / (T[j] - T[m]);
// python's math.prod starts from the int 1, so the first factor is taken as is.
__device__ __forceinline__ void lagrange4(float t, float c[4]) {
  const double Td[4] = {0.0, 1.0 / 3.0, 2.0 / 3.0, 1.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float p = 1.0f;
    bool first = true;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m == j) continue;
      // torch: (t_f32 - python_double) -> fp32 op with the scalar rounded to fp32; same for the divide
      float f = (t - (float)Td[m]) / (float)(Td[j] - Td[m]);
      p = first ? f : p * f;
      first = false;
    }
    c[j] = p;
  }
}

// Time-slice pair and blend weights of HashGridT.forward (model/hash_field.py:79-85).
struct SlicePair {
  int i1, i2;
  float w1, w2;
};
__device__ __forceinline__ SlicePair slice_pair(float t, int n_slices) {
  SlicePair s;
  float idx = t * (float)(n_slices - 1);
  float f = floorf(idx), c = ceilf(idx);
  s.i1 = (int)f;
  s.i2 = (int)c;
  if (s.i1 == s.i2) {
    s.w1 = 1.0f;
    s.w2 = 0.0f;
  } else {
    s.w1 = c - idx;
    s.w2 = idx - f;
  }
  return s;
}
