// Ray sampling, volumetric compositing (forward + backward), mask compaction and the small per-sample
// glue of the renderer for gfx950.  Replaces the torch op chains of model/renderer.py:59-129 and the
// masked gather / scatter of model/lidar4d.py:196-219, plus tcnn's Frequency encoding (lidar4d.py:68-74).
//
// Compositing runs one 64-lane wave per ray over chunks of 64 consecutive samples (lane = sample, dense accesses): the
// wave combines the lanes' (1 - alpha) with a scan and carries the transmittance from chunk to chunk; the
// `weights > 1e-4` test is compacted with ballots (wave-level compaction): lanes write their surviving sample indices
// into a slot range reserved by ONE atomicAdd per ray.
#include "common.h"

#define MAXC 16  // samples per lane per segment -> segments of 1024 samples

__device__ __forceinline__ float wave_excl_scan_mul(float v, int lane, float& total) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(inc, d, 64);
    if (lane >= d) inc *= o;
  }
  total = __shfl(inc, 63, 64);
  float ex = __shfl_up(inc, 1, 64);
  return lane == 0 ? 1.0f : ex;
}

__device__ __forceinline__ float wave_excl_scan_add(float v, int lane, float& total) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  total = __shfl(inc, 63, 64);
  float ex = __shfl_up(inc, 1, 64);
  return lane == 0 ? 0.0f : ex;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ---- sampling (renderer.py:77-89) ---------------------------------------------------------------
__global__ void __launch_bounds__(256) sample_rays_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ lin, const float* __restrict__ noise,
                                                         int64_t N, int T, float near, float far, float bound,
                                                         float* __restrict__ z_vals, float* __restrict__ xyz) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * T) return;
  const int64_t ray = idx / T;
  const int t = (int)(idx - ray * T);
  float z = near + (far - near) * lin[t];
  if (noise) {
    const float sample_dist = (far - near) / (float)T;
    z = z + (noise[idx] - 0.5f) * sample_dist;
  }
  z_vals[idx] = z;
  if (xyz) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v = rays_o[ray * 3 + k] + rays_d[ray * 3 + k] * z;
      v = fminf(fmaxf(v, -bound), bound);
      xyz[idx * 3 + k] = v;
    }
  }
}

// ---- compositing forward (renderer.py:98-110,121-126) ------------------------------------------
__device__ __forceinline__ float alpha_of(float delta, float sigma, float density_scale, int active) {
  // 1 - exp(-deltas * density_scale * sigma)  /  1 - exp(-2 * deltas * density_scale * sigma)
  float e = active ? ((-2.0f * delta) * density_scale) * sigma : ((-delta) * density_scale) * sigma;
  return 1.0f - expf(e);
}

#define CF_RAYS 16  // rays (wavefronts) per workgroup of composite_fwd_kernel
__global__ void __launch_bounds__(64 * CF_RAYS) composite_fwd_kernel(const float* __restrict__ sigma, const float* __restrict__ z_vals,
                                                           int64_t N, int T, float sample_dist, float density_scale,
                                                           int active, float* __restrict__ weights,
                                                           float* __restrict__ weights_sum, float* __restrict__ depth,
                                                           uint8_t* __restrict__ mask, int32_t* __restrict__ mask_idx,
                                                           int32_t* __restrict__ mask_count) {
  // The work list's slots are reserved per WORKGROUP (16 rays): one returning atomic per ray on the one counter was 16,384
  // same-address atomics per launch, served one after the other -- 0.2 ms whatever the rest of the kernel did.
  __shared__ int wave_keep[CF_RAYS], block_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray_raw = (int64_t)blockIdx.x * CF_RAYS + wave;
  const bool live = ray_raw < N;                 // (wave-uniform; a wavefront behind the last ray walks the last ray and stores nothing:
  const int64_t ray = live ? ray_raw : N - 1;    //  every wavefront has to reach the barriers below)
  const float* sg = sigma + ray * T;
  const float* zv = z_vals + ray * T;
  float carry = 1.0f;  // transmittance entering the chunk
  float wsum = 0.0f, dsum = 0.0f;
  // Chunks of 64 CONSECUTIVE samples, lane = sample: every load and store is one dense 256-byte access.  (Lane-owned
  // runs of T/64 samples made each instruction touch 24 cache lines that were refetched for every k: the backward
  // kernel fetched 4 GB for 0.3 GB of inputs, profiles/r01_pmc_FETCH_SIZE_c3_v10.txt.)
  for (int seg0 = 0; seg0 < T; seg0 += 64 * MAXC) {
    const int seg_len = min(T - seg0, 64 * MAXC);
    const int c = (seg_len + 63) / 64;  // chunks in this segment
    unsigned long long keep[MAXC];
    int n_keep = 0;
    // ALL of the segment's loads first: one wavefront walks a ray's 12 chunks in order (the transmittance is carried from chunk to
    // chunk), and with a load -> scan -> store chain per chunk it paid 12 memory latencies in a row -- 0.23 ms for 150 MB of
    // traffic (profiles/r04_floor_table.md: HBM floor 0.03 ms).
    float zs[MAXC], zn[MAXC], sgs[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      zs[k] = 0.0f; zn[k] = 0.0f; sgs[k] = 0.0f;
      if (k >= c) continue;  // wave-uniform
      const int j = min(seg0 + k * 64 + lane, T - 1);  // (lanes past the end read the last sample and ignore it: no divergent load)
      zs[k] = zv[j];
      zn[k] = zv[min(j + 1, T - 1)];
      sgs[k] = sg[j];
    }
    // ... then all the arithmetic (the weights stay in registers), then all the stores: a store between two chunks made the
    // compiler's conservative s_waitcnt in front of the next chunk's first operand wait for that store's acknowledgement.
    float ws[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      keep[k] = 0ull;
      ws[k] = 0.0f;
      if (k >= c) continue;  // wave-uniform
      const int j = seg0 + k * 64 + lane;
      const bool in = j < seg0 + seg_len;
      float a = 0.0f, f = 1.0f, z = 0.0f;
      if (in) {
        z = zs[k];
        const float delta = (j + 1 < T) ? (zn[k] - z) : sample_dist;
        a = alpha_of(delta, sgs[k], density_scale, active);
        f = (1.0f - a) + 1e-15f;
      }
      float total;
      const float tr = carry * wave_excl_scan_mul(f, lane, total);
      carry *= total;
      const float w = a * tr;
      bool m = false;
      if (in) {
        ws[k] = w;
        wsum += w;
        dsum += w * z;
        m = live && w > 1e-4f;
      }
      keep[k] = __ballot(m);
      n_keep += __popcll(keep[k]);
    }
    // compaction: the workgroup's wavefronts post their counts, ONE atomicAdd reserves the slots of all of them (requested before
    // the stores below, consumed behind them), every wavefront takes its share in ray order
    if (mask_idx) {  // (uniform)
      if (lane == 0) wave_keep[wave] = n_keep;
      __syncthreads();
      if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < CF_RAYS; ++w) tot += wave_keep[w];
        block_base = tot > 0 ? atomicAdd(mask_count, tot) : 0;
      }
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < MAXC; ++k) {
        if (k >= c) continue;  // wave-uniform
        const int j = seg0 + k * 64 + lane;
        if (j < seg0 + seg_len) {
          weights[ray * T + j] = ws[k];
          if (mask) mask[ray * T + j] = (keep[k] >> lane) & 1ull ? 1 : 0;
        }
      }
    }
    if (mask_idx) {
      __syncthreads();
      int base = block_base;
      for (int w = 0; w < wave; ++w) base += wave_keep[w];  // (wave-uniform)
      if (n_keep > 0) {
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
          if (k >= c) continue;
          const unsigned long long below = keep[k] & ((1ull << lane) - 1ull);
          if ((keep[k] >> lane) & 1ull) mask_idx[base + __popcll(below)] = (int32_t)(ray * T + seg0 + k * 64 + lane);
          base += __popcll(keep[k]);
        }
      }
      __syncthreads();  // (wave_keep / block_base are rewritten by the next segment of a long ray)
    }
  }
  wsum = wave_sum(wsum);
  dsum = wave_sum(dsum);
  if (lane == 0 && live) {
    if (weights_sum) weights_sum[ray] = wsum;
    if (depth) depth[ray] = dsum;
  }
}

// image[ray][c] = sum_t weights * attr   (renderer.py:129)
__global__ void __launch_bounds__(256) composite_image_kernel(const float* __restrict__ weights, const float* __restrict__ attr,
                                                             int64_t N, int T, int C, float* __restrict__ image) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= N) return;
  for (int c = 0; c < C; ++c) {
    float s = 0.0f;
    for (int j = lane; j < T; j += 64) s += weights[ray * T + j] * attr[(ray * T + j) * C + c];
    s = wave_sum(s);
    if (lane == 0) image[ray * C + c] = s;
  }
}

// ---- compositing backward ------------------------------------------------------------------------
// gw_i = d_depth z_i + d_wsum + sum_c d_image_c attr_ic + d_weights_i
// w_i = a_i T_i, T_i = prod_{j<i} (1 - a_j + eps):  dL/da_i = gw_i T_i - (sum_{k>i} gw_k w_k) / (1 - a_i + eps)
// da_i/dsigma_i = kappa delta_i density_scale (1 - a_i),  kappa = 2 if active_sensor else 1
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ sigma, const float* __restrict__ z_vals,
                                                           const float* __restrict__ weights, const float* __restrict__ attr,
                                                           int64_t N, int T, int C, float sample_dist, float density_scale,
                                                           int active, const float* __restrict__ d_depth,
                                                           const float* __restrict__ d_wsum, const float* __restrict__ d_image,
                                                           const float* __restrict__ d_weights, float* __restrict__ d_sigma,
                                                           float* __restrict__ d_attr) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= N) return;
  const float* sg = sigma + ray * T;
  const float* zv = z_vals + ray * T;
  const float gd = d_depth ? d_depth[ray] : 0.0f;
  const float gs = d_wsum ? d_wsum[ray] : 0.0f;
  float gi[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C && c < 4; ++c) gi[c] = d_image ? d_image[ray * C + c] : 0.0f;
  const float kappa = active ? 2.0f : 1.0f;
  // Same chunking as the forward kernel (lane = sample inside a chunk of 64 consecutive samples: dense accesses).
  // Segments are walked from the far end so the suffix sum is available; the transmittance entering a segment comes
  // from a first pass over the segment products, the one entering a chunk from a forward pass inside the segment.
  const int nseg = (T + 64 * MAXC - 1) / (64 * MAXC);
  float seg_in[8];  // supports T <= 8192
  {
    float carry = 1.0f;
    for (int s = 0; s < nseg && s < 8; ++s) {
      seg_in[s] = carry;
      if (s + 1 >= nseg) break;  // the last segment's product is nobody's input (T <= 1024: this pass does not run at all)
      const int seg0 = s * 64 * MAXC, seg_len = min(T - seg0, 64 * MAXC), c = (seg_len + 63) / 64;
      float prod = 1.0f;
      for (int k = 0; k < c; ++k) {
        const int j = seg0 + k * 64 + lane;
        if (j < seg0 + seg_len) {
          const float z = zv[j];
          const float delta = (j + 1 < T) ? (zv[j + 1] - z) : sample_dist;
          prod *= (1.0f - alpha_of(delta, sg[j], density_scale, active)) + 1e-15f;
        }
      }
      float total;
      (void)wave_excl_scan_mul(prod, lane, total);
      carry *= total;
    }
  }
  const int nc = min(C, 4);
  float suffix_carry = 0.0f;  // sum_{k in later chunks / segments} gw_k w_k
  for (int s = nseg - 1; s >= 0; --s) {
    const int seg0 = s * 64 * MAXC, seg_len = min(T - seg0, 64 * MAXC), c = (seg_len + 63) / 64;
    // Three phases, as in the forward kernel: every load of the segment (no arithmetic on the way, or the loads would be waited
    // for one by one), then the arithmetic on registers, then every store -- one memory latency per segment instead of two per
    // chunk (0.27 ms for 0.35 GB of traffic before).  The sums are formed in the order they always were: bit-identical results.
    float zs[MAXC], dl[MAXC], a[MAXC], wv[MAXC];  // a[]: sigma as loaded, then alpha, then d_sigma
    float2_t at[MAXC];  // (one register pair per load, split where it is used: separate arrays are copies behind the load)
    const bool two = attr && nc == 2 && C == 2;  // the LiDAR head (ray-drop, intensity): its attribute rows are hoisted too
    // (inputs that do not exist are read from a place that does and never used: a conditional load merges with its default in a
    // register copy, and the copy waits for the load)
    const float* atp = two ? attr : weights;
    const int64_t atm = two ? 2 : 0;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      zs[k] = 0.0f; dl[k] = 0.0f; a[k] = 0.0f; wv[k] = 0.0f; at[k] = float2_t{0.0f, 0.0f};
      if (k >= c) continue;  // wave-uniform
      const int j = min(seg0 + k * 64 + lane, T - 1);  // (lanes past the end read the last sample and ignore it: no divergent load)
      zs[k] = zv[j];
      dl[k] = zv[min(j + 1, T - 1)];
      a[k] = sg[j];
      wv[k] = weights[ray * T + j];
      at[k] = *reinterpret_cast<const float2_t*>(atp + (ray * T + j) * atm);
    }
    float trv[MAXC];
    float carry = seg_in[s];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {  // forward: alpha and the transmittance of every sample
      trv[k] = 0.0f;
      if (k >= c) continue;
      const int j = seg0 + k * 64 + lane;
      float f = 1.0f, al = 0.0f, d = 0.0f;
      if (j < seg0 + seg_len) {
        d = (j + 1 < T) ? (dl[k] - zs[k]) : sample_dist;
        al = alpha_of(d, a[k], density_scale, active);
        f = (1.0f - al) + 1e-15f;
      }
      dl[k] = d;
      a[k] = al;
      float total;
      trv[k] = carry * wave_excl_scan_mul(f, lane, total);
      carry *= total;
    }
#pragma unroll
    for (int k = MAXC - 1; k >= 0; --k) {  // backward: suffix sums
      if (k >= c) continue;
      const int j = seg0 + k * 64 + lane;
      const bool in = j < seg0 + seg_len;
      float g = 0.0f, w = 0.0f;
      if (in) {
        w = wv[k];
        g = gd * zs[k] + gs;
        if (d_weights) g += d_weights[ray * T + j];  // (a gradient of the weights themselves: not on the training path, loaded here)
        if (two) {
          g += gi[0] * at[k][0];
          g += gi[1] * at[k][1];
        } else if (attr) {  // other widths (not on the LiDAR path): loaded here
          for (int cc = 0; cc < nc; ++cc) g += gi[cc] * attr[(ray * T + j) * C + cc];
        }
      }
      const float q = g * w;
      float qtotal;
      const float before = wave_excl_scan_add(q, lane, qtotal);
      const float suf = (qtotal - before - q) + suffix_carry;  // samples after this one
      float ds = 0.0f;
      if (in) {
        const float one_m = (1.0f - a[k]) + 1e-15f;
        const float da = g * trv[k] - suf / one_m;
        ds = da * (kappa * dl[k] * density_scale * (1.0f - a[k]));
      }
      a[k] = ds;
      suffix_carry += qtotal;
    }
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      if (k >= c) continue;
      const int j = seg0 + k * 64 + lane;
      if (j < seg0 + seg_len) {
        d_sigma[ray * T + j] = a[k];
        if (d_attr) {
          float* dp = d_attr + (ray * T + j) * C;
          if (nc == 2 && C == 2) *reinterpret_cast<float2_t*>(dp) = float2_t{wv[k] * gi[0], wv[k] * gi[1]};
          else
            for (int cc = 0; cc < nc; ++cc) dp[cc] = wv[k] * gi[cc];
        }
      }
    }
  }
}

// ---- tcnn Frequency encoding (lidar4d.py:68-74) ------------------------------------
// exact range reduction: y = x * 2^k is exact in fp32, y mod 2 is exact, then sinpi/cospi.
__global__ void __launch_bounds__(256) freq_kernel(const float* __restrict__ x, int64_t P, int n_dims, int n_freq,
                                                  half_t* __restrict__ out, int out_stride) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = n_dims * n_freq;
  if (idx >= P * per_row) return;
  const int64_t p = idx / per_row;
  const int q = (int)(idx - p * per_row);
  const int d = q / n_freq, k = q - d * n_freq;
  const float y = ldexpf(x[p * n_dims + d], k);
  const float r = y - 2.0f * floorf(y * 0.5f);
  half2_t v;
  v[0] = f2h(sinpif(r));
  v[1] = f2h(cospif(r));
  *reinterpret_cast<half2_t*>(out + p * out_stride + 2 * q) = v;
}

// ---- attribute() glue (lidar4d.py:196-219) -------------------------------------------------------
// gather: X_attr[j] = [dir_enc[ray(j)] (n_enc) | geo_feat[idx[j]] (n_geo, = h[:, 1:1+n_geo]) | ones] as fp16
__global__ void __launch_bounds__(256) attr_gather_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                                                         int64_t cap, int T, const half_t* __restrict__ dir_enc, int n_enc,
                                                         const half_t* __restrict__ h, int n_geo, half_t* __restrict__ xa,
                                                         int in_pad) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = in_pad / 8;
  const int64_t j = gid / chunks;
  const int ch = (int)(gid - j * chunks);
  const int64_t M = count ? min((int64_t)*count, cap) : cap;
  if (j >= M) return;
  const int64_t p = idx ? idx[j] : j;
  const int64_t ray = p / T;
  if ((n_enc & 7) == 0 && ch * 8 + 8 <= n_enc) {  // chunk inside the direction encoding: one aligned 16-byte copy
    *reinterpret_cast<uint4*>(xa + j * in_pad + ch * 8) = *reinterpret_cast<const uint4*>(dir_enc + ray * n_enc + ch * 8);
    return;
  }
  half_t hrow[16];  // the sample's sigma-net output row: geo_feat = columns 1 .. n_geo
  *reinterpret_cast<uint4*>(hrow) = *reinterpret_cast<const uint4*>(h + p * 16);
  *reinterpret_cast<uint4*>(hrow + 8) = *reinterpret_cast<const uint4*>(h + p * 16 + 8);
  half_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = ch * 8 + e;
    half_t val = (half_t)1.0f;
    if (col < n_enc) val = dir_enc[ray * n_enc + col];
    else if (col < n_enc + n_geo) {
      const int g = 1 + (col - n_enc);
      half_t t = hrow[0];
#pragma unroll
      for (int q = 1; q < 16; ++q) t = (g == q) ? hrow[q] : t;  // register select: no dynamic indexing
      val = t;
    }
    v[e] = val;
  }
  *reinterpret_cast<uint4*>(xa + j * in_pad + ch * 8) = *reinterpret_cast<uint4*>(v);
}

// scatter: attr[idx[j]] = sigmoid(fp16 MLP outputs) rounded to fp16 (lidar4d.py:210-219);
// channel 0 = ray-drop, channel 1 = intensity
__global__ void __launch_bounds__(256) attr_scatter_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                                                          int64_t cap, const half_t* __restrict__ y_raydrop,
                                                          const half_t* __restrict__ y_intensity, float* __restrict__ attr,
                                                          float* __restrict__ attr_compact) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t M = count ? min((int64_t)*count, cap) : cap;
  if (j >= M) return;
  const int64_t p = idx ? idx[j] : j;
  const float r = h2f(f2h(1.0f / (1.0f + expf(-h2f(y_raydrop[j * 16])))));
  const float it = h2f(f2h(1.0f / (1.0f + expf(-h2f(y_intensity[j * 16])))));
  if (attr) {
    attr[p * 2 + 0] = r;
    attr[p * 2 + 1] = it;
  }
  if (attr_compact) {
    attr_compact[j * 2 + 0] = r;
    attr_compact[j * 2 + 1] = it;
  }
}

// backward of the scatter + sigmoid: dy[j][0] = d_attr[idx[j]][c] * s (1 - s) * loss_scale, other columns 0
__global__ void __launch_bounds__(256) attr_scatter_bwd_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                                                              int64_t cap, const float* __restrict__ d_attr,
                                                              const float* __restrict__ attr_compact, float loss_scale,
                                                              half_t* __restrict__ dy_raydrop, half_t* __restrict__ dy_intensity) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t M = count ? min((int64_t)*count, cap) : cap;
  if (j >= M) return;
  const int64_t p = idx ? idx[j] : j;
  const float sr = attr_compact[j * 2 + 0], si = attr_compact[j * 2 + 1];
  half_t vr[16], vi[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { vr[e] = (half_t)0.0f; vi[e] = (half_t)0.0f; }
  vr[0] = f2h_grad(d_attr[p * 2 + 0] * sr * (1.0f - sr) * loss_scale);
  vi[0] = f2h_grad(d_attr[p * 2 + 1] * si * (1.0f - si) * loss_scale);
  uint4* dr = reinterpret_cast<uint4*>(dy_raydrop + j * 16);
  uint4* di = reinterpret_cast<uint4*>(dy_intensity + j * 16);
  dr[0] = reinterpret_cast<uint4*>(vr)[0]; dr[1] = reinterpret_cast<uint4*>(vr)[1];
  di[0] = reinterpret_cast<uint4*>(vi)[0]; di[1] = reinterpret_cast<uint4*>(vi)[1];
}

// backward of the gather: dh[idx[j]][1 + k] += dxa_raydrop[j][n_enc + k] + dxa_intensity[j][n_enc + k]
// (values are already loss-scaled fp16; dh is fp16, same scale).  Each sample appears once in idx: plain stores.
__global__ void __launch_bounds__(256) attr_gather_bwd_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                                                             int64_t cap, const half_t* __restrict__ dxa_r,
                                                             const half_t* __restrict__ dxa_i, int in_pad, int n_enc,
                                                             int n_geo, half_t* __restrict__ dh) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t j = gid / n_geo;
  const int k = (int)(gid - j * n_geo);
  const int64_t M = count ? min((int64_t)*count, cap) : cap;
  if (j >= M) return;
  const int64_t p = idx ? idx[j] : j;
  const float v = h2f(dxa_r[j * in_pad + n_enc + k]) + h2f(dxa_i[j * in_pad + n_enc + k]);
  dh[p * 16 + 1 + k] = f2h_grad(v);
}

// Same, one thread per row with 16-byte accesses, for the usual layout (n_enc a multiple of 8, 16 columns available):
// reads columns n_enc .. n_enc+15 of both gradients, writes the whole 32-byte dh row (column 0, the sigma gradient, is
// written afterwards by sigma_bwd_kernel; rows outside idx keep their zeros).
__global__ void __launch_bounds__(256) attr_gather_bwd_rows_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ count,
                                                                  int64_t cap, const half_t* __restrict__ dxa_r,
                                                                  const half_t* __restrict__ dxa_i, int in_pad, int n_enc,
                                                                  int n_geo, half_t* __restrict__ dh, int h_layout) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t M = count ? min((int64_t)*count, cap) : cap;
  if (j >= M) return;
  const int64_t p = idx ? idx[j] : j;
  half_t r[16], q[16], o[16];
  *reinterpret_cast<uint4*>(r) = *reinterpret_cast<const uint4*>(dxa_r + j * in_pad + n_enc);
  *reinterpret_cast<uint4*>(r + 8) = *reinterpret_cast<const uint4*>(dxa_r + j * in_pad + n_enc + 8);
  *reinterpret_cast<uint4*>(q) = *reinterpret_cast<const uint4*>(dxa_i + j * in_pad + n_enc);
  *reinterpret_cast<uint4*>(q + 8) = *reinterpret_cast<const uint4*>(dxa_i + j * in_pad + n_enc + 8);
  o[0] = (half_t)0.0f;
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    // h_layout: the 16 columns read already have the sigma-network row's order [-, g0 .. g14] (l4d_attr_mlp_bwd)
    const float v = h_layout ? h2f(r[1 + k]) + h2f(q[1 + k]) : h2f(r[k]) + h2f(q[k]);
    o[1 + k] = k < n_geo ? f2h_grad(v) : (half_t)0.0f;
  }
  *reinterpret_cast<uint4*>(dh + p * 16) = *reinterpret_cast<uint4*>(o);
  *reinterpret_cast<uint4*>(dh + p * 16 + 8) = *reinterpret_cast<uint4*>(o + 8);
}

// sigma = trunc_exp(h[:,0]) (activation.py:6-20) and its backward into dh[:,0] (fp16, loss-scaled)
__global__ void __launch_bounds__(256) sigma_from_h_kernel(const half_t* __restrict__ h, int64_t P, float* __restrict__ sigma) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  sigma[p] = expf(h2f(h[p * 16]));
}
__global__ void __launch_bounds__(256) sigma_bwd_kernel(const half_t* __restrict__ h, const float* __restrict__ d_sigma, int64_t P,
                                                       float loss_scale, half_t* __restrict__ dh) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float x = fminf(fmaxf(h2f(h[p * 16]), -15.0f), 15.0f);
  const float g = d_sigma[p] * expf(x) * loss_scale;
  dh[p * 16] = f2h_grad(g);
}

// dh[p] = [d_sigma[p] * exp(clamp(h0, -15, 15)) * loss_scale, 0 x 15] as WHOLE 32-byte rows, from sigma = exp(h0) itself (exp is
// monotone: clamping h0 to [-15, 15] is clamping sigma to [exp(-15), exp(15)], with the same expf at both ends) -- the zero fill of
// dh and sigma_bwd_kernel's strided 2-byte read / write of every row in one dense pass.  The attribute networks' backward then
// adds its 15 geo-feature columns and keeps column 0 (l4d_attr_mlp_bwd_gathered, dh_accumulate bit 1).
__global__ void __launch_bounds__(256) sigma_bwd_rows_kernel(const float* __restrict__ sigma, const float* __restrict__ d_sigma, int64_t P,
                                                            float loss_scale, half_t* __restrict__ dh) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float e = fminf(fmaxf(sigma[p], expf(-15.0f)), expf(15.0f));
  const half_t g = f2h_grad(d_sigma[p] * e * loss_scale);
  uint4 lo = make_uint4((uint32_t)__builtin_bit_cast(unsigned short, g), 0u, 0u, 0u);
  uint4* dst = reinterpret_cast<uint4*>(dh + p * 16);
  dst[0] = lo;
  dst[1] = make_uint4(0u, 0u, 0u, 0u);
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int l4d_sigma_bwd_rows(const float* sigma, const float* d_sigma, int64_t P, float loss_scale, void* dh, void* stream) {
  if (P == 0) return 0;
  L4D_LAUNCH(sigma_bwd_rows_kernel, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream, sigma, d_sigma, P, loss_scale,
             (half_t*)dh);
  L4D_LAUNCH_CHECK("l4d_sigma_bwd_rows");
  return 0;
}

extern "C" int l4d_sample_rays(const float* rays_o, const float* rays_d, const float* lin, const float* noise, int64_t N,
                               int32_t T, float near, float far, float bound, float* z_vals, float* xyz, void* stream) {
  if (N == 0) return 0;
  L4D_LAUNCH(sample_rays_kernel, dim3((unsigned)ceil_div64(N * T, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                     rays_d, lin, noise, N, T, near, far, bound, z_vals, xyz);
  L4D_LAUNCH_CHECK("l4d_sample_rays");
  return 0;
}

extern "C" int l4d_composite_fwd(const float* sigma, const float* z_vals, int64_t N, int32_t T, float sample_dist,
                                 float density_scale, int32_t active_sensor, float* weights, float* weights_sum,
                                 float* depth, uint8_t* mask, int32_t* mask_idx, int32_t* mask_count, void* stream) {
  if (N == 0) return 0;
  if (mask_count) {
    l4d_fill_async(mask_count, 0u, sizeof(int32_t), (hipStream_t)stream);
  }
  L4D_LAUNCH(composite_fwd_kernel, dim3((unsigned)ceil_div64(N, CF_RAYS)), dim3(64 * CF_RAYS), 0, (hipStream_t)stream, sigma, z_vals,
                     N, T, sample_dist, density_scale, active_sensor, weights, weights_sum, depth, mask, mask_idx, mask_count);
  L4D_LAUNCH_CHECK("l4d_composite_fwd");
  return 0;
}

extern "C" int l4d_composite_image(const float* weights, const float* attr, int64_t N, int32_t T, int32_t C, float* image,
                                   void* stream) {
  if (N == 0) return 0;
  L4D_LAUNCH(composite_image_kernel, dim3((unsigned)ceil_div64(N, 4)), dim3(256), 0, (hipStream_t)stream, weights, attr,
                     N, T, C, image);
  L4D_LAUNCH_CHECK("l4d_composite_image");
  return 0;
}

extern "C" int l4d_composite_bwd(const float* sigma, const float* z_vals, const float* weights, const float* attr, int64_t N,
                                 int32_t T, int32_t C, float sample_dist, float density_scale, int32_t active_sensor,
                                 const float* d_depth, const float* d_wsum, const float* d_image, const float* d_weights,
                                 float* d_sigma, float* d_attr, void* stream) {
  if (N == 0) return 0;
  if (C > 4 || T > 8 * 64 * MAXC) { l4d_set_error(1, "l4d_composite_bwd: C <= 4 and T <= 8192"); return 1; }
  L4D_LAUNCH(composite_bwd_kernel, dim3((unsigned)ceil_div64(N, 4)), dim3(256), 0, (hipStream_t)stream, sigma, z_vals,
                     weights, attr, N, T, C, sample_dist, density_scale, active_sensor, d_depth, d_wsum, d_image, d_weights,
                     d_sigma, d_attr);
  L4D_LAUNCH_CHECK("l4d_composite_bwd");
  return 0;
}

extern "C" int l4d_freq_fwd(const float* x, int64_t P, int32_t n_dims, int32_t n_freq, void* out, int32_t out_stride,
                            void* stream) {
  if (P == 0) return 0;
  L4D_LAUNCH(freq_kernel, dim3((unsigned)ceil_div64(P * n_dims * n_freq, 256)), dim3(256), 0, (hipStream_t)stream, x, P,
                     n_dims, n_freq, (half_t*)out, out_stride);
  L4D_LAUNCH_CHECK("l4d_freq_fwd");
  return 0;
}

extern "C" int l4d_attr_gather(const int32_t* idx, const int32_t* count, int64_t cap, int32_t T, const void* dir_enc,
                               int32_t n_enc, const void* h, int32_t n_geo, void* xa, int32_t in_pad, void* stream) {
  if (cap == 0) return 0;
  if (in_pad % 8 || n_enc + n_geo > in_pad || n_geo > 15) { l4d_set_error(1, "l4d_attr_gather: bad widths"); return 1; }
  L4D_LAUNCH(attr_gather_kernel, dim3((unsigned)ceil_div64(cap * (in_pad / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     idx, count, cap, T, (const half_t*)dir_enc, n_enc, (const half_t*)h, n_geo, (half_t*)xa, in_pad);
  L4D_LAUNCH_CHECK("l4d_attr_gather");
  return 0;
}

extern "C" int l4d_attr_scatter(const int32_t* idx, const int32_t* count, int64_t cap, const void* y_raydrop,
                                const void* y_intensity, float* attr, float* attr_compact, void* stream) {
  if (cap == 0) return 0;
  L4D_LAUNCH(attr_scatter_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, (hipStream_t)stream, idx, count,
                     cap, (const half_t*)y_raydrop, (const half_t*)y_intensity, attr, attr_compact);
  L4D_LAUNCH_CHECK("l4d_attr_scatter");
  return 0;
}

extern "C" int l4d_attr_scatter_bwd(const int32_t* idx, const int32_t* count, int64_t cap, const float* d_attr,
                                    const float* attr_compact, float loss_scale, void* dy_raydrop, void* dy_intensity,
                                    void* stream) {
  if (cap == 0) return 0;
  L4D_LAUNCH(attr_scatter_bwd_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, (hipStream_t)stream, idx,
                     count, cap, d_attr, attr_compact, loss_scale, (half_t*)dy_raydrop, (half_t*)dy_intensity);
  L4D_LAUNCH_CHECK("l4d_attr_scatter_bwd");
  return 0;
}

extern "C" int l4d_attr_gather_bwd(const int32_t* idx, const int32_t* count, int64_t cap, const void* dxa_raydrop,
                                   const void* dxa_intensity, int32_t in_pad, int32_t n_enc, int32_t n_geo, void* dh,
                                   int32_t h_layout, void* stream) {
  if (cap == 0) return 0;
  if (h_layout && !((n_enc & 7) == 0 && n_enc + 16 <= in_pad && n_geo <= 15)) {
    l4d_set_error(1, "l4d_attr_gather_bwd: h_layout needs 16 aligned columns at n_enc");
    return 1;
  }
  if ((n_enc & 7) == 0 && n_enc + 16 <= in_pad && n_geo <= 15)
    L4D_LAUNCH(attr_gather_bwd_rows_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, count, cap, (const half_t*)dxa_raydrop, (const half_t*)dxa_intensity, in_pad, n_enc, n_geo,
                       (half_t*)dh, h_layout);
  else
    L4D_LAUNCH(attr_gather_bwd_kernel, dim3((unsigned)ceil_div64(cap * n_geo, 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, count, cap, (const half_t*)dxa_raydrop, (const half_t*)dxa_intensity, in_pad, n_enc, n_geo,
                       (half_t*)dh);
  L4D_LAUNCH_CHECK("l4d_attr_gather_bwd");
  return 0;
}

extern "C" int l4d_sigma_from_h(const void* h, int64_t P, float* sigma, void* stream) {
  if (P == 0) return 0;
  L4D_LAUNCH(sigma_from_h_kernel, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)h, P, sigma);
  L4D_LAUNCH_CHECK("l4d_sigma_from_h");
  return 0;
}

extern "C" int l4d_sigma_bwd(const void* h, const float* d_sigma, int64_t P, float loss_scale, void* dh, void* stream) {
  if (P == 0) return 0;
  L4D_LAUNCH(sigma_bwd_kernel, dim3((unsigned)ceil_div64(P, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)h,
                     d_sigma, P, loss_scale, (half_t*)dh);
  L4D_LAUNCH_CHECK("l4d_sigma_bwd");
  return 0;
}
