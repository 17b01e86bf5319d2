// Optimiser step and precision casts for gfx950: pure HBM-streaming kernels, 16 bytes per lane.
// Adam follows torch.optim.Adam as configured by the reference (main_lidar4d.py:298-300: betas
// (0.9, 0.99), eps 1e-15, no weight decay, no amsgrad) and emits the fp16 compute copy of the
// parameters in the same pass (tiny-cuda-nn casts its fp32 master parameters to fp16 every forward).
#include "common.h"

__global__ void __launch_bounds__(256) cast_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int64_t n) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const float4_t v = *reinterpret_cast<const float4_t*>(src + i4);
    half4_t h;
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = f2h(v[k]);
    *reinterpret_cast<half4_t*>(dst + i4) = h;
  } else {
    for (int64_t i = i4; i < n; ++i) dst[i] = f2h(src[i]);
  }
}

// param/grad/moments fp32.  grad_scale multiplies the gradient first (1/loss_scale, 1/world_size ...).
// bias_c1 = 1 - beta1^t, bias_c2 = 1 - beta2^t.  Update order follows torch's single-tensor Adam:
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2; denom = sqrt(v)/sqrt(bias_c2) + eps; p -= (lr/bias_c1) m/denom
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                  float* __restrict__ m, float* __restrict__ v, half_t* __restrict__ p16,
                                                  int64_t n, float lr, float b1, float b2, float eps, float bias_c1,
                                                  float bias_c2, float grad_scale) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const float step = lr / bias_c1;
  const float rs2 = sqrtf(bias_c2);
  if (i4 + 3 < n) {
    float4_t p = *reinterpret_cast<float4_t*>(param + i4);
    const float4_t g4 = *reinterpret_cast<const float4_t*>(grad + i4);
    float4_t mm = *reinterpret_cast<float4_t*>(m + i4);
    float4_t vv = *reinterpret_cast<float4_t*>(v + i4);
    half4_t h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = g4[k] * grad_scale;
      mm[k] = mm[k] + (g - mm[k]) * (1.0f - b1);
      vv[k] = vv[k] * b2 + (1.0f - b2) * g * g;
      const float denom = sqrtf(vv[k]) / rs2 + eps;
      p[k] = p[k] - step * (mm[k] / denom);
      h[k] = f2h(p[k]);
    }
    *reinterpret_cast<float4_t*>(param + i4) = p;
    *reinterpret_cast<float4_t*>(m + i4) = mm;
    *reinterpret_cast<float4_t*>(v + i4) = vv;
    if (p16) *reinterpret_cast<half4_t*>(p16 + i4) = h;
  } else {
    for (int64_t i = i4; i < n; ++i) {
      const float g = grad[i] * grad_scale;
      m[i] = m[i] + (g - m[i]) * (1.0f - b1);
      v[i] = v[i] * b2 + (1.0f - b2) * g * g;
      const float denom = sqrtf(v[i]) / rs2 + eps;
      param[i] = param[i] - step * (m[i] / denom);
      if (p16) p16[i] = f2h(param[i]);
    }
  }
}

extern "C" int l4d_cast_f32_to_f16(const float* src, void* dst, int64_t n, void* stream) {
  if (n == 0) return 0;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) { l4d_set_error(1, "l4d_cast: src must be 16-byte, dst 8-byte aligned"); return 1; }
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)ceil_div64(ceil_div64(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (half_t*)dst, n);
  L4D_LAUNCH_CHECK("l4d_cast_f32_to_f16");
  return 0;
}

extern "C" int l4d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_f16, int64_t n,
                             float lr, float beta1, float beta2, float eps, float bias_c1, float bias_c2, float grad_scale,
                             void* stream) {
  if (n == 0) return 0;
  if (((uintptr_t)param & 15) || ((uintptr_t)grad & 15) || ((uintptr_t)exp_avg & 15) || ((uintptr_t)exp_avg_sq & 15) ||
      ((uintptr_t)param_f16 & 7)) {
    l4d_set_error(1, "l4d_adam_step: buffers must be 16-byte aligned (fp16 copy 8-byte)");
    return 1;
  }
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ceil_div64(ceil_div64(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, (half_t*)param_f16, n, lr, beta1, beta2, eps, bias_c1, bias_c2, grad_scale);
  L4D_LAUNCH_CHECK("l4d_adam_step");
  return 0;
}
