// Optimiser step and precision casts for gfx950: pure HBM-streaming kernels, 16 bytes per lane.
// Adam follows torch.optim.Adam as configured by the reference (main_lidar4d.py:298-300: betas
// (0.9, 0.99), eps 1e-15, no weight decay, no amsgrad) and emits the fp16 compute copy of the
// parameters in the same pass (tiny-cuda-nn casts its fp32 master parameters to fp16 every forward).
#include "common.h"
#include "wave_dev.h"
#include <algorithm>

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

// ---- ranged / gated Adam with device-side state ------------------------------------------------------------------
// The reference's optimiser state is per PARAMETER TENSOR: torch.optim.Adam skips tensors whose .grad is None (no moment
// decay, no step increment) -- which is what happens to the 6-7 of 8 HashGridT time slices a step does not touch
// (hash_field.py:79-85 under zero_grad(set_to_none)) -- and torch.cuda.amp.GradScaler skips the WHOLE step when a
// gradient is non-finite (runner.py:506-508).  Both decisions are taken on the device here, so the step never waits on
// the host: every range r of the flat arena has a step counter steps[r] and, optionally, a gate gates[gate_idx[r]]
// (a float the backward pass sets to non-zero when it produced a gradient for the range; it lives in the gradient
// arena's tail so a data-parallel SUM all-reduce merges the ranks' gates for free); scaler[2] != 0 = "gradients were
// non-finite: skip everything", scaler[3] = 1 / loss scale.
#define ADAM_MAX_RANGES 40
struct AdamRanges {
  int n;
  int64_t off[ADAM_MAX_RANGES], len[ADAM_MAX_RANGES];
  float lr[ADAM_MAX_RANGES];
  int gate[ADAM_MAX_RANGES];  // index into gates, or -1: always on
};

__device__ __forceinline__ bool range_on(const AdamRanges& R, int r, const float* gates, const float* scaler) {
  if (scaler && scaler[2] != 0.0f) return false;
  return R.gate[r] < 0 || gates[R.gate[r]] != 0.0f;
}

// sched (optional, 2 floats on the device): [0] scheduler iterations so far, [1] the learning-rate factor of THIS step.  The
// reference steps a LambdaLR every iteration, skipped or not (main_lidar4d.py:303-305: 0.1 ** min(it / iters, 1)); with the
// schedule on the device a captured step (hipGraph) needs no kernel argument that changes from replay to replay.
__global__ void adam_advance_kernel(AdamRanges R, const float* __restrict__ gates, const float* __restrict__ scaler,
                                    int32_t* __restrict__ steps, float* __restrict__ sched, float sched_iters) {
  const int r = threadIdx.x;
  if (r < R.n && range_on(R, r, gates, scaler)) steps[r] += 1;
  if (r == 0 && sched) {
    const double it = (double)sched[0];
    sched[1] = (float)pow(0.1, fmin(it / (double)sched_iters, 1.0));
    sched[0] = (float)(it + 1.0);
  }
}

__global__ void __launch_bounds__(256) adam_ranges_kernel(AdamRanges R, float* __restrict__ param, const float* __restrict__ grad,
                                                         float* __restrict__ m, float* __restrict__ v, half_t* __restrict__ p16,
                                                         const float* __restrict__ gates, const float* __restrict__ scaler,
                                                         const int32_t* __restrict__ steps, float b1, float b2, float eps,
                                                         float grad_scale, const float* __restrict__ sched) {
  const int r = blockIdx.y;
  const int64_t n = R.len[r];
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if ((int64_t)blockIdx.x * blockDim.x * 4 >= n) return;   // block-uniform
  if (!range_on(R, r, gates, scaler)) return;              // launch-uniform per range
  __shared__ float s_step, s_rs2;
  if (threadIdx.x == 0) {  // bias corrections as torch computes them (python doubles), from the range's own step count
    const double t = (double)steps[r];
    const double lr = (double)R.lr[r] * (sched ? (double)sched[1] : 1.0);
    s_step = (float)(lr / (1.0 - pow((double)b1, t)));
    s_rs2 = (float)sqrt(1.0 - pow((double)b2, t));
  }
  __syncthreads();
  if (i4 >= n) return;
  const float step = s_step, rs2 = s_rs2;
  const float gs = grad_scale * (scaler ? scaler[3] : 1.0f);
  const int64_t o = R.off[r] + i4;
  if (i4 + 3 < n) {
    float4_t p = *reinterpret_cast<float4_t*>(param + o);
    const float4_t g4 = *reinterpret_cast<const float4_t*>(grad + o);
    float4_t mm = *reinterpret_cast<float4_t*>(m + o);
    float4_t vv = *reinterpret_cast<float4_t*>(v + o);
    half4_t h;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = g4[k] * gs;
      mm[k] = mm[k] + (g - mm[k]) * (1.0f - b1);
      vv[k] = vv[k] * b2 + (1.0f - b2) * g * g;
      const float denom = sqrtf(vv[k]) / rs2 + eps;
      p[k] = p[k] - step * (mm[k] / denom);
      h[k] = f2h(p[k]);
    }
    *reinterpret_cast<float4_t*>(param + o) = p;
    *reinterpret_cast<float4_t*>(m + o) = mm;
    *reinterpret_cast<float4_t*>(v + o) = vv;
    if (p16) *reinterpret_cast<half4_t*>(p16 + o) = h;
  } else {
    for (int64_t i = o; i < R.off[r] + n; ++i) {
      const float g = grad[i] * gs;
      m[i] = m[i] + (g - m[i]) * (1.0f - b1);
      v[i] = v[i] * b2 + (1.0f - b2) * g * g;
      const float denom = sqrtf(v[i]) / rs2 + eps;
      param[i] = param[i] - step * (m[i] / denom);
      if (p16) p16[i] = f2h(param[i]);
    }
  }
}

// scaler state (4 floats): [0] loss scale S, [1] growth tracker, [2] found non-finite (this step), [3] 1 / S
__global__ void __launch_bounds__(256) nonfinite_check_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ state) {
  bool bad = false;
  for (int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < n; i4 += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i4 + 3 < n) {
      const float4_t v = *reinterpret_cast<const float4_t*>(g + i4);
#pragma unroll
      for (int k = 0; k < 4; ++k) bad |= !(fabsf(v[k]) <= 3.402823466e38f);
    } else {
      for (int64_t i = i4; i < n; ++i) bad |= !(fabsf(g[i]) <= 3.402823466e38f);
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.0f;  // benign race: everybody stores the same value
}

// max |x| over n floats into out[0] (zeroed by the caller's fill); any inf / nan makes it +inf.
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  float m = 0.0f;
  bool bad = false;
  for (int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i4 < n; i4 += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i4 + 3 < n) {
      const float4_t v = *reinterpret_cast<const float4_t*>(x + i4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { m = fmaxf(m, fabsf(v[k])); bad |= !(fabsf(v[k]) <= 3.402823466e38f); }
    } else {
      for (int64_t i = i4; i < n; ++i) { m = fmaxf(m, fabsf(x[i])); bad |= !(fabsf(x[i]) <= 3.402823466e38f); }
    }
  }
  m = __any(bad) ? __builtin_inff() : wave_max(m);
  // one atomic per WORKGROUP: all waves of a launch finish within microseconds of each other, and a few thousand atomics on one
  // address serialise (measured: 139 us for an 8 MB input with one atomic per wave)
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (m > 0.0f) atomic_max_nonneg(out, m);
  }
}

// torch.cuda.amp.GradScaler.update(): halve after a non-finite step, double after growth_interval clean steps
__global__ void scaler_update_kernel(float* __restrict__ state, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (state[2] != 0.0f) {
    state[0] *= backoff;
    state[1] = 0.0f;
  } else {
    state[1] += 1.0f;
    if (state[1] >= (float)interval) {
      state[0] *= growth;
      state[1] = 0.0f;
    }
  }
  state[2] = 0.0f;
  state[3] = 1.0f / state[0];
}

// gates[i1] = gates[i2] = 1 for the time-slice pair HashGridT.forward selects at tinfo[0] (hash_field.py:79-85)
__global__ void mark_slices_kernel(const float* __restrict__ tinfo, int n_slices, float* __restrict__ gates) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const SlicePair sp = slice_pair(tinfo[0], n_slices);
  gates[sp.i1] = 1.0f;
  gates[sp.i2] = 1.0f;
}

extern "C" int l4d_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_f16,
                                    int32_t n_ranges, const int64_t* off, const int64_t* len, const float* lr,
                                    const int32_t* gate_idx, const float* gates, const float* scaler, int32_t* steps,
                                    float beta1, float beta2, float eps, float grad_scale, float* sched, float sched_iters,
                                    void* stream) {
  if (n_ranges == 0) return 0;
  if (n_ranges > ADAM_MAX_RANGES) { l4d_set_error(1, "l4d_adam_step_ranges: too many ranges"); return 1; }
  AdamRanges R;
  R.n = n_ranges;
  int64_t max_len = 0;
  for (int r = 0; r < n_ranges; ++r) {
    R.off[r] = off[r]; R.len[r] = len[r]; R.lr[r] = lr[r]; R.gate[r] = gate_idx ? gate_idx[r] : -1;
    if (off[r] & 3) { l4d_set_error(1, "l4d_adam_step_ranges: range offsets must be multiples of 4 elements"); return 1; }
    if (R.gate[r] >= 0 && !gates) { l4d_set_error(1, "l4d_adam_step_ranges: gated range without a gate buffer"); return 1; }
    max_len = std::max(max_len, len[r]);
  }
  if (((uintptr_t)param & 15) || ((uintptr_t)grad & 15) || ((uintptr_t)exp_avg & 15) || ((uintptr_t)exp_avg_sq & 15) ||
      ((uintptr_t)param_f16 & 7)) {
    l4d_set_error(1, "l4d_adam_step_ranges: buffers must be 16-byte aligned (fp16 copy 8-byte)");
    return 1;
  }
  if (max_len == 0) return 0;
  L4D_LAUNCH(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, R, gates, scaler, steps, sched, sched_iters);
  L4D_LAUNCH(adam_ranges_kernel, dim3((unsigned)ceil_div64(ceil_div64(max_len, 4), 256), n_ranges), dim3(256), 0,
                     (hipStream_t)stream, R, param, grad, exp_avg, exp_avg_sq, (half_t*)param_f16, gates, scaler, steps, beta1,
                     beta2, eps, grad_scale, (const float*)sched);
  L4D_LAUNCH_CHECK("l4d_adam_step_ranges");
  return 0;
}

extern "C" int l4d_grad_nonfinite_check(const float* grad, int64_t n, float* scaler_state, void* stream) {
  if (n == 0) return 0;
  if ((uintptr_t)grad & 15) { l4d_set_error(1, "l4d_grad_nonfinite_check: grad must be 16-byte aligned"); return 1; }
  const int64_t blocks = std::min<int64_t>(2048, ceil_div64(ceil_div64(n, 4), 256));
  L4D_LAUNCH(nonfinite_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, n, scaler_state);
  L4D_LAUNCH_CHECK("l4d_grad_nonfinite_check");
  return 0;
}

extern "C" int l4d_absmax_f32(const float* x, int64_t n, float* out, void* stream) {
  l4d_fill_async(out, 0u, sizeof(float), (hipStream_t)stream);
  if (n == 0) return 0;
  if ((uintptr_t)x & 15) { l4d_set_error(1, "l4d_absmax_f32: x must be 16-byte aligned"); return 1; }
  const int64_t blocks = std::min<int64_t>(256, ceil_div64(ceil_div64(n, 4), 256));
  L4D_LAUNCH(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
  L4D_LAUNCH_CHECK("l4d_absmax_f32");
  return 0;
}

extern "C" int l4d_scaler_update(float* scaler_state, float growth_factor, float backoff_factor, int32_t growth_interval,
                                 void* stream) {
  L4D_LAUNCH(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler_state, growth_factor,
                     backoff_factor, growth_interval);
  L4D_LAUNCH_CHECK("l4d_scaler_update");
  return 0;
}

extern "C" int l4d_mark_time_slices(const float* tinfo, int32_t n_slices, float* gates, void* stream) {
  L4D_LAUNCH(mark_slices_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tinfo, n_slices, gates);
  L4D_LAUNCH_CHECK("l4d_mark_time_slices");
  return 0;
}

extern "C" int l4d_cast_f32_to_f16(const float* src, void* dst, int64_t n, void* stream) {
  if (n == 0) return 0;
  if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) { l4d_set_error(1, "l4d_cast: src must be 16-byte, dst 8-byte aligned"); return 1; }
  L4D_LAUNCH(cast_kernel, dim3((unsigned)ceil_div64(ceil_div64(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
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
  L4D_LAUNCH(adam_kernel, dim3((unsigned)ceil_div64(ceil_div64(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, (half_t*)param_f16, n, lr, beta1, beta2, eps, bias_c1, bias_c2, grad_scale);
  L4D_LAUNCH_CHECK("l4d_adam_step");
  return 0;
}
