/*
 * lidar4d_hip.h -- C ABI of the MI355X-native LiDAR4D ray-rendering hot path (gfx950 / CDNA4).
 *
 * Drop-in boundary.  The reference binds this path to native code through two Python-level
 * operator surfaces, and this header is what a maintainer would bind in their place
 * (INTEGRATION.md shows the ctypes stubs):
 *
 *   - tiny-cuda-nn's `tcnn.Encoding` / `tcnn.Network` modules (reference call sites
 *     model/hash_field.py:47-57,107-117; model/flow_field.py:67-77; model/lidar4d.py:68-117)
 *       -> l4d_hashgrid_*, l4d_hashgrid_t_*, l4d_freq_fwd, l4d_mlp_*
 *   - ATen kernels reached from the reference's torch code on the path
 *     (F.grid_sample model/planes_field.py:77-81; cumprod/exp/mask model/renderer.py:77-129;
 *      masked gather/scatter model/lidar4d.py:196-219)
 *       -> l4d_planes_*, l4d_sample_rays, l4d_composite_*, l4d_attr_*
 *
 * Conventions (same as the reference's only in-tree native op, utils/chamfer3D/chamfer3D.cu:135-194:
 * caller-allocated outputs, int status): every pointer is a DEVICE pointer unless its comment says
 * "host"; tensors are dense row-major; `stream` is a hipStream_t passed as void*; every entry point
 * returns 0 on success or a hipError_t value (l4d_last_error() gives the text).  No torch types.
 * fp16 buffers are IEEE binary16 (`_Float16`), passed as void*.
 */
#ifndef LIDAR4D_HIP_H
#define LIDAR4D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L4D_MAX_LEVELS 16
#define L4D_ABI_VERSION 2

/* Multi-resolution hash grid geometry (tiny-cuda-nn "HashGrid" encoding).
 * Filled by host code (lidar4d_amd/gridmeta.py) exactly as tiny-cuda-nn's host code derives it. */
typedef struct {
  int32_t n_dims;      /* 2 or 3 */
  int32_t n_features;  /* features per level: 2, 4 or 8 */
  int32_t n_levels;    /* <= L4D_MAX_LEVELS */
  uint32_t hashed_mask; /* bit l set: level l uses the coherent-prime hash, else dense strides */
  float scale[L4D_MAX_LEVELS];
  uint32_t res[L4D_MAX_LEVELS];
  uint32_t size[L4D_MAX_LEVELS];   /* entries in level */
  uint32_t offset[L4D_MAX_LEVELS]; /* first entry of level, in entries */
} l4d_grid_desc;

int l4d_version(void);
const char* l4d_last_error(void);

/* Per-kernel timing for benchmarks: while enabled, every kernel the library launches is bracketed by a pair of HIP events
 * on its launch stream.  enable(1) clears earlier records and starts recording, enable(0) stops and clears;
 * l4d_profile_get(i, &name, &ms) waits for record i and returns the kernel's name (as at the launch site, template
 * arguments included) and duration.  Single host thread. */
int l4d_profile_enable(int32_t on);
int l4d_profile_count(void);
int l4d_profile_get(int32_t i, const char** name /*host out*/, float* ms /*host out*/);

/* Side streams.  Entry points that launch several independent kernels (l4d_density_encode_fwd / _bwd) fork them onto
 * library-owned side streams of the launch stream and join them back before they return (event record + wait only: no host
 * synchronisation, capturable into a hipGraph).  No reference counterpart (PyTorch runs the path on one stream).
 *   l4d_streams_config(mask)  bit 0: forward encode (the xz / yz LDS evaluation next to the plane columns), bit 1: field
 *                             adjoint (sorted scatter | time planes | static planes + dynamic hash), bit 2: the static grid's
 *                             level-major pre-pass next to the LDS evaluation of the xz / yz stacks.  Default 0 = everything
 *                             on the launch stream (environment L4D_STREAMS overrides): measured on MI355X the overlap buys
 *                             0 +- 0.4 ms of a 40 ms step -- these kernels share their bottlenecks.
 *   l4d_streams_mask()        current setting
 *   l4d_streams_join(stream)  `stream` waits for all outstanding side-stream work
 *   l4d_side_fork(from, i) / l4d_side_join(into, i): side stream i (0..2) continues from the end of `from` and is returned
 *                             (null on failure) / `into` waits for it -- for callers that overlap their own launches. */
int l4d_streams_config(int32_t mask);
int l4d_streams_mask(void);
int l4d_streams_join(void* stream);
void* l4d_side_fork(void* from, int32_t i);
int l4d_side_join(void* into, int32_t i);

/* ---- tcnn.Encoding(HashGrid) : model/hash_field.py:107-117, model/flow_field.py:67-77 -------
 * x        [P, x_stride] fp32; the grid's n_dims coordinates are columns cols[0..n_dims-1] (host)
 * table    [n_entries, F] fp16 (level-major)
 * out      [P, out_stride] fp16, columns [0, L*F)
 */
int l4d_hashgrid_fwd(const l4d_grid_desc* desc /*host*/, const float* x, int64_t P, int32_t x_stride,
                     const int32_t* cols /*host*/, const void* table, void* out, int32_t out_stride,
                     void* stream);
/* The same with l4d_hashgrid_fwd_workspace() bytes of device scratch: the levels are evaluated one after the other over the whole
 * chip (every XCD's L2 then holds the ONE table in use instead of all of them), x-neighbour entry pairs in one 16-byte load where
 * they share an aligned pair (F = 4, 16-byte aligned table), level-major into the scratch, and a second streaming kernel writes
 * the rows.  Worthwhile from ~1e6 points with tables that exceed an L2 (4 MB).
 * With x_stride == 4 and cols = (0, 1, 2) (rows [x, y, z, t]) the level-major kernels read a point's row as ONE 16-byte load:
 * x must then hold 4 * P readable floats (the fourth one of every row is loaded and ignored); l4d_hashgrid_t_fwd_ws likewise. */
int64_t l4d_hashgrid_fwd_workspace(const l4d_grid_desc* desc /*host*/, int64_t P);
int l4d_hashgrid_fwd_ws(const l4d_grid_desc* desc /*host*/, const float* x, int64_t P, int32_t x_stride,
                        const int32_t* cols /*host*/, const void* table, void* out, int32_t out_stride, void* workspace,
                        void* stream);
/* dout [P, dout_stride] fp16 (dout_is_half=1) or fp32; grad_table [n_entries, F] fp32, ACCUMULATED
 * into (atomics); gradient is multiplied by grad_scale before accumulation. */
int l4d_hashgrid_bwd(const l4d_grid_desc* desc /*host*/, const float* x, int64_t P, int32_t x_stride,
                     const int32_t* cols /*host*/, const void* dout, int32_t dout_stride,
                     int32_t dout_is_half, float grad_scale, float* grad_table, void* stream);

/* ---- HashGridT.forward : model/hash_field.py:76-88; FlowField grid + interpT : flow_field.py:102-125 ---
 * Time blend of two slices + cubic-Lagrange interpT over the 4 feature chunks of each level, fused.
 * tables   host array of n_slices device pointers, each a [n_entries, F] fp16 table (n_slices = 1: no blend)
 * t        device pointer to ONE fp32: the call's time in [0,1] (no host sync: slice pair, blend
 *          weights and Lagrange coefficients are derived on the device)
 * out      [P, out_stride] fp32 or fp16 (out_is_half), columns [0, L*F/4); F in {4, 8}
 */
int l4d_hashgrid_t_fwd(const l4d_grid_desc* desc /*host*/, const float* x, int64_t P, int32_t x_stride,
                       const int32_t* cols /*host*/, const void* const* tables /*host*/, int32_t n_slices,
                       const float* t, void* out, int32_t out_stride, int32_t out_is_half, void* stream);
/* The same with l4d_hashgrid_t_fwd_workspace() bytes of device scratch: the levels are evaluated one after the other over the whole
 * chip (every L2 then holds ONE level's table), level-major into the scratch, and a streaming kernel writes the rows.  Used for the
 * flow field's grid (3-D, F = 8, fp16 rows) from 2^18 points on; every other shape takes l4d_hashgrid_t_fwd. */
int64_t l4d_hashgrid_t_fwd_workspace(const l4d_grid_desc* desc /*host*/, int64_t P);
int l4d_hashgrid_t_fwd_ws(const l4d_grid_desc* desc /*host*/, const float* x, int64_t P, int32_t x_stride,
                          const int32_t* cols /*host*/, const void* const* tables /*host*/, int32_t n_slices,
                          const float* t, void* out, int32_t out_stride, int32_t out_is_half, void* workspace, void* stream);
/* dout [P, dout_stride] fp32 or fp16, multiplied by grad_scale; grad_tables host array of n_slices device
 * pointers to fp32 tables (accumulated into).  Only the slices selected by *t are touched.
 * scratch: n_entries * F/4 floats of device memory (the per-entry scalar accumulators; zeroed here).
 * workspace: null, or l4d_hashgrid_t_bwd_workspace() bytes -> the hashed levels use the sorted (binned) scatter
 * instead of global atomics (needs fp16 dout); worthwhile from ~1e6 points. */
int64_t l4d_hashgrid_t_bwd_workspace(const l4d_grid_desc* desc /*host*/, int64_t P);
int l4d_hashgrid_t_bwd(const l4d_grid_desc* desc /*host*/, const float* x, int64_t P, int32_t x_stride,
                       const int32_t* cols /*host*/, int32_t n_slices, const float* t, const void* dout,
                       int32_t dout_stride, int32_t dout_is_half, float grad_scale,
                       float* const* grad_tables /*host*/, float* scratch, void* workspace, void* stream);

/* ---- Planes4D : model/planes_field.py:87-141,198-239 --------------------------------------------
 * planes_cl  channel-last copy of the hex-plane parameters: for scale s, plane c (comb order
 *            (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)) a [H, W, C] fp32 block at element offset plane_off[s*6+c]
 *            (host array), H = res of comb[1], W = res of comb[0].
 * res        host [n_scales*4] resolutions (x,y,z,t) per scale
 * xt         [P, 4] fp32
 * which      0 = both, 1 = static only, 2 = dynamic only
 * out_s/out_d [P, n_scales*C] fp32 (may be null when not requested)
 */
int l4d_planes_relayout(const float* const* planes /*host: n_scales*6 device ptrs, [1,C,H,W]*/,
                        const int32_t* res /*host*/, int32_t n_scales, int32_t C, float* planes_cl,
                        const int64_t* plane_off /*host*/,
                        int32_t to_channel_last /*1: planes -> planes_cl; 0: planes_cl -> planes; 2: planes += planes_cl*/,
                        void* stream);
int l4d_planes_fwd(const float* planes_cl, const int64_t* plane_off /*host*/, const int32_t* res /*host*/,
                   int32_t n_scales, int32_t C, const float* xt, int64_t P, int32_t which, float* out_s,
                   float* out_d, void* stream);
/* grad_cl accumulated (channel-last layout, atomics); dxt [P,4] fp32 overwritten when non-null
 * (ATen grid_sampler border rule: zero where the un-normalised coordinate was clipped). */
int l4d_planes_bwd(const float* planes_cl, const int64_t* plane_off /*host*/, const int32_t* res /*host*/,
                   int32_t n_scales, int32_t C, const float* xt, int64_t P, int32_t which,
                   const float* dout_s, const float* dout_d, float* grad_cl, float* dxt, void* stream);

/* ---- tcnn.Encoding(Frequency) : model/lidar4d.py:68-74,208 --------------------------------------
 * x [P, 3] fp32 -> out [P, out_stride] fp16 cols [0, 72): per dim [sin(2^k pi x), cos(2^k pi x)] k<12 */
int l4d_freq_fwd(const float* x, int64_t P, int32_t n_dims, int32_t n_freq, void* out, int32_t out_stride,
                 void* stream);

/* ---- tcnn.Network(FullyFusedMLP) : model/lidar4d.py:83-117, nn.Linear stack flow_field.py:84-98 ---
 * Bias-free ReLU MLP on MFMA (v_mfma_f32_16x16x32_f16, fp32 accumulate, fp16 activations).
 * x        [P, in_pad] fp16 (caller pads with 1.0; in_pad multiple of 16, <= 128)
 * weights  fp16, layers concatenated: [64, in_pad], (n_hidden-1) x [64, 64], [16, 64]
 * y        [P, 16] fp16
 * act      optional [n_hidden, P, 64] fp16 hidden activations (saved for backward) or null
 * n_rows   optional device int32: only the first min(P, *n_rows) rows are processed (work lists sized
 *          on the device, e.g. the weights>1e-4 compaction); buffers keep their P-row strides */
int l4d_mlp_fwd(const void* x, int64_t P, const int32_t* n_rows, int32_t in_pad, int32_t n_hidden,
                const void* weights, void* y, void* act, void* stream);
/* l4d_mlp_fwd with the density activation as epilogue: additionally sigma[row] = exp(y[row][0]) (fp32; trunc_exp's
 * forward, activation.py:6-20).  Same (in_pad, n_hidden) support as l4d_mlp_fwd. */
int l4d_mlp_fwd_sigma(const void* x, int64_t P, int32_t in_pad, int32_t n_hidden, const void* weights, void* y, void* act,
                      float* sigma, void* stream);
/* dy [P,16] fp16 (already multiplied by the caller's loss scale); dx [P, in_pad] fp16 or null;
 * grad_w fp32, same layout as weights, ACCUMULATED with 1/loss_scale applied.
 * act: the forward's saved activations, or null = recompute them from x inside the kernel (saves 128 B/row/layer of HBM
 * traffic twice; built for in_pad <= 32 with 1-3 hidden layers and for the 128 -> 64 -> 16 density network, where it is the
 * default of the fused path since round 5: -0.17 ms per step; other shapes would spill registers and return an error).
 * dx_absmax: null, or a device fp32 that receives (atomic max; zero it first) the largest |dx| of the input columns
 * [absmax_col_lo, absmax_col_hi) (multiples of 16) as the kernel stores them, +inf if one of them is not finite -- the
 * consumer of those columns then needs no pass of its own to scale its fixed-point accumulators (l4d_density_encode_bwd). */
int l4d_mlp_bwd(const void* x, const void* act, const void* dy, int64_t P, const int32_t* n_rows, int32_t in_pad,
                int32_t n_hidden, const void* weights, void* dx, float* grad_w, float inv_loss_scale, float* dx_absmax,
                int32_t absmax_col_lo, int32_t absmax_col_hi, void* stream);

/* ---- LiDAR_Renderer.run : model/renderer.py:59-129 -----------------------------------------------
 * sample: lin [T] = torch.linspace(0,1,T) (renderer.py:77; passed in so that the bits are the caller's
 * torch build's), noise [N,T] in [0,1) or null (renderer.py:84) -> z_vals [N,T]; xyz [N*T,3] clipped to
 * [-bound,bound] (renderer.py:88-89) or null */
int l4d_sample_rays(const float* rays_o, const float* rays_d, const float* lin, const float* noise, int64_t N,
                    int32_t T, float near, float far, float bound, float* z_vals, float* xyz, void* stream);
/* composite: sigma [N,T] -> weights [N,T], weights_sum [N], depth [N] (= sum w z, not normalised),
 * mask [N,T] (uint8, weights > 1e-4, renderer.py:110) and the compacted list of masked flat sample
 * indices (grouped by ray, ascending inside a ray) + its length (device int32, zeroed here). */
int l4d_composite_fwd(const float* sigma, const float* z_vals, int64_t N, int32_t T, float sample_dist,
                      float density_scale, int32_t active_sensor, float* weights, float* weights_sum,
                      float* depth, uint8_t* mask, int32_t* mask_idx, int32_t* mask_count, void* stream);
/* image [N,C] = sum_t weights * attr [N*T, C]  (renderer.py:129) */
int l4d_composite_image(const float* weights, const float* attr, int64_t N, int32_t T, int32_t C,
                        float* image, void* stream);
/* backward of depth = sum w z, wsum = sum w, image = sum w attr (and optional direct d_weights)
 * wrt sigma [N,T] and attr [N*T,C] (d_attr may be null) */
int l4d_composite_bwd(const float* sigma, const float* z_vals, const float* weights, const float* attr,
                      int64_t N, int32_t T, int32_t C, float sample_dist, float density_scale,
                      int32_t active_sensor, const float* d_depth, const float* d_wsum, const float* d_image,
                      const float* d_weights /*[N,T] or null*/, float* d_sigma, float* d_attr, void* stream);

/* ---- LiDAR4D.attribute glue : model/lidar4d.py:196-219 -------------------------------------------
 * Work-list form of `x[mask]` / `output[mask] = h`: idx [cap] int32 flat sample indices (null = identity),
 * count = device int32 with the list length (null = cap); no host sync is needed to size the launch.
 * gather : xa[j] = [dir_enc[idx[j] / T] (n_enc fp16) | h[idx[j]][1..n_geo] (geo_feat, fp16) | 1.0 pad]
 * scatter: attr[idx[j]] = (sigmoid(y_raydrop[j][0]), sigmoid(y_intensity[j][0])) rounded to fp16;
 *          attr [P,2] fp32 must be zero-filled by the caller; attr_compact [cap,2] keeps the same values
 * *_bwd  : the two adjoints; dy_* [cap,16] fp16 and dh [P,16] fp16 carry loss_scale */
int l4d_attr_gather(const int32_t* idx, const int32_t* count, int64_t cap, int32_t T, const void* dir_enc,
                    int32_t n_enc, const void* h, int32_t n_geo, void* xa, int32_t in_pad, void* stream);
int l4d_attr_scatter(const int32_t* idx, const int32_t* count, int64_t cap, const void* y_raydrop,
                     const void* y_intensity, float* attr, float* attr_compact, void* stream);
int l4d_attr_scatter_bwd(const int32_t* idx, const int32_t* count, int64_t cap, const float* d_attr,
                         const float* attr_compact, float loss_scale, void* dy_raydrop, void* dy_intensity,
                         void* stream);
/* dh [P,16] fp16 (zero-filled by the caller): columns 1..n_geo of row idx[j] <- dxa_raydrop[j] + dxa_intensity[j] (geo
 * columns); column 0 of those rows is written as 0 -- call l4d_sigma_bwd AFTER this to fill it. */
int l4d_attr_gather_bwd(const int32_t* idx, const int32_t* count, int64_t cap, const void* dxa_raydrop,
                        const void* dxa_intensity, int32_t in_pad, int32_t n_enc, int32_t n_geo, void* dh,
                        int32_t h_layout /*0; 1: columns n_enc .. n_enc + 15 are ordered [-, g0 .. g14] (l4d_attr_mlp_bwd)*/,
                        void* stream);
/* The attribute networks straight on the work list.  l4d_attr_mlp_fwd = l4d_mlp_fwd whose input row j is assembled in the kernel
 * from dir_enc[idx[j] / T] (n_enc columns), the sigma network's output row h[idx[j]] and ones, instead of reading the
 * [cap, in_pad] matrix l4d_attr_gather would have to write first.  in_pad = 96, n_enc in {64, 72, 80}, n_geo = 15.  The 16
 * columns from n_enc on are taken in the order of the sigma network's row -- [1.0, g0 .. g14], two aligned 16-byte loads -- and
 * the weight columns are permuted to match ("physical" column order).  x_rows_out: null, or [cap, in_pad] fp16 that receives the
 * assembled rows in physical order for l4d_attr_mlp_bwd (one network of the pair stores them, both read them).
 * l4d_attr_mlp_bwd: x_rows as stored by the forward; dx_tail [cap, in_pad - 64] fp16 <- the input gradient's columns 64 ..
 * in_pad - 1 in physical order (the direction encoding has no trainable input): feed it to l4d_attr_gather_bwd with
 * in_pad = in_pad - 64, n_enc = n_enc - 64, h_layout = 1.
 * Sigmoid epilogue (lidar4d.py:210-219): with attr_dense [P, 2] fp32 (pre-zeroed) and attr_compact [cap, 2] fp32 given,
 * s = fp16(sigmoid(y[:, 0])) goes to attr_dense[idx[j]][channel] and attr_compact[j][channel] (channel 0 = ray-drop, 1 =
 * intensity) and y is not stored (may be null): replaces l4d_attr_scatter. */
int l4d_attr_mlp_fwd(const int32_t* idx, const int32_t* count, int64_t cap, int32_t T, const void* dir_enc, int32_t n_enc,
                     const void* h, int32_t n_geo, int32_t in_pad, int32_t n_hidden, const void* weights, void* y, void* act,
                     void* x_rows_out, float* attr_dense, float* attr_compact, int32_t channel, void* stream);
int l4d_attr_mlp_bwd(const void* x_rows, const int32_t* count, int64_t cap, int32_t n_enc, int32_t n_geo, int32_t in_pad,
                     int32_t n_hidden, const void* act, const void* dy, const void* weights, void* dx_tail, float* grad_w,
                     float inv_loss_scale, void* stream);
/* The same backward with the rows assembled from (idx, dir_enc, h) as in the forward, a tile ahead of their use: nothing is
 * stored by the forward (x_rows_out null).  n_hidden 1 or 2 (three hidden layers: use the stored-rows form).
 * With d_attr given (dy / dx_tail then unused, may be null) the streaming steps either side of the network run inside the
 * kernel (lidar4d.py:210-219 backwards): dy[j][0] = d_attr[idx[j]][channel] * s (1 - s) * loss_scale with
 * s = attr_compact[j][channel] (replaces l4d_attr_scatter_bwd), and the geo-feature gradient goes straight to
 * dh[idx[j]][1 .. 15] (fp16 [samples, 16], pre-zeroed; dh_accumulate bit 0: stored (0) / added (1); bit 1: column 0 of dh is kept
 * (it already holds the density activation's adjoint, l4d_sigma_bwd_rows) instead of zeroed; added to what the other network of the
 * pair stored when 1; replaces l4d_attr_gather_bwd). */
int l4d_attr_mlp_bwd_gathered(const int32_t* idx, const int32_t* count, int64_t cap, int32_t T, const void* dir_enc,
                              int32_t n_enc, const void* h, int32_t n_geo, int32_t in_pad, int32_t n_hidden, const void* act,
                              const void* dy, const void* weights, void* dx_tail, float* grad_w, float inv_loss_scale,
                              const float* d_attr, const float* attr_compact, int32_t channel, float loss_scale, void* dh,
                              int32_t dh_accumulate, void* stream);
/* sigma = trunc_exp(h[:,0]) (model/activation.py:6-20) on the sigma net's fp16 output h [P,16], and its
 * adjoint dh[:,0] = d_sigma * exp(clamp(h0,-15,15)) * loss_scale (fp16) */
int l4d_sigma_from_h(const void* h, int64_t P, float* sigma, void* stream);
int l4d_sigma_bwd(const void* h, const float* d_sigma, int64_t P, float loss_scale, void* dh, void* stream);
/* The same adjoint as whole rows: dh[p] = [d_sigma[p] * exp(clamp(h0, -15, 15)) * loss_scale, 0 x 15] with exp(h0) = sigma[p] taken
 * from the forward's sigma (fp32 [P]); replaces the zero fill of dh + l4d_sigma_bwd when it runs BEFORE the attribute networks'
 * backward, which then keeps column 0 (l4d_attr_mlp_bwd_gathered: dh_accumulate bit 1). */
int l4d_sigma_bwd_rows(const float* sigma, const float* d_sigma, int64_t P, float loss_scale, void* dh, void* stream);

/* ---- LiDAR4D.density, fused field evaluation : model/lidar4d.py:139-179 ---------------------------
 * One launch evaluates, per sample point, the hex-planes at (x,t) and at the two flow-warped neighbour
 * frames, the static hash grid, the three HashGridT stacks at the three frames, the 0.5/0.25/0.25 blends and
 * the concat, and writes the sigma network's padded fp16 input row.  Compute copies of the parameters: */
#define L4D_MAX_TIME_SLICES 8
#define L4D_MAX_PLANE_SCALES 8
typedef struct {
  l4d_grid_desc hash_static;                                   /* 3-D, F = 4 */
  const void* hash_static_table;                               /* fp16 */
  l4d_grid_desc hash_dynamic[3];                               /* xy, xz, yz: 2-D, F = 4 */
  const void* hash_dynamic_tables[3][L4D_MAX_TIME_SLICES];     /* fp16, one table per time slice */
  const void* hash_dynamic_pairs[3];                           /* null, or l4d_dyn_pairs_build() of the plane's slice tables */
  int32_t n_slices;
  int32_t n_scales;                                            /* hex-plane scales */
  int32_t plane_channels;                                      /* 8 */
  int32_t plane_res[L4D_MAX_PLANE_SCALES * 4];                 /* (x,y,z,t) resolution per scale */
  int64_t plane_off[L4D_MAX_PLANE_SCALES * 6];                 /* element offsets into planes_cl */
  const float* planes_cl;                                      /* channel-last fp32 planes */
} l4d_field_desc;
typedef struct {                                               /* fp32 gradient buffers, accumulated into */
  float* hash_static_table;
  float* hash_dynamic_tables[3][L4D_MAX_TIME_SLICES];
  float* planes_cl;
} l4d_field_grads;

int l4d_field_width(const l4d_field_desc* f /*host*/);         /* feature columns before padding */
/* Pair-interleaved copy of one plane's time-slice tables for the fused forward: pairs [n_slices - 1][n_entries][2][4] fp16,
 * pairs[q][e] = {slice q entry e, slice q + 1 entry e}.  HashGridT blends two adjacent slices (hash_field.py:79-85): with
 * this copy both come in ONE 16-byte load per corner.  slice_tables: host array of n_slices device pointers ([n_entries, 4]
 * fp16 each, all levels).  Rebuild whenever the fp16 tables change (one pass over ~50 MB per plane set: microseconds). */
int l4d_dyn_pairs_build(const void* const* slice_tables /*host*/, int32_t n_slices, int64_t n_entries, void* pairs, void* stream);
/* tinfo [8] fp32 (device) <- t: [t, t1=(f+1)/num_frames, t2=(f-1)/num_frames, has_fwd, has_bwd, f] */
int l4d_time_setup(const float* t, int32_t num_frames, float* tinfo, void* stream);
/* l4d_sample_rays variant that writes xt [N*T,4] = ((clip(o+d z)+bound)/(2 bound), t) (lidar4d.py:141,148-149) */
int l4d_sample_rays_xt(const float* rays_o, const float* rays_d, const float* lin, const float* noise,
                       const float* t, int64_t N, int32_t T, float near, float far, float bound, float* z_vals,
                       float* xt, void* stream);
/* flow16 [P,16] fp16: flow network output (cols 0-2 forward, 3-5 backward); X [P,in_pad] fp16.
 * hd_scratch: null, or l4d_density_encode_fwd_workspace() bytes of device scratch -> the xz / yz 2-D x time hash stacks are
 * evaluated by a separate kernel from LDS-resident slice tables (faster from ~1e5 samples), and -- from 2^18 samples on, with
 * plane_rows given as well -- the static 3-D grid by a level-major pre-pass into the same scratch (one level's table at a time
 * over the whole chip), whose columns the encode kernel then reads instead of gathering.
 * plane_rows: null, or l4d_plane_rows_workspace() bytes of device scratch -> the time planes are first reduced to the
 * 1-D rows of the call's three frame times (their time coordinate is the same for every sample) and sampled with two
 * taps instead of four. */
int64_t l4d_plane_rows_workspace(const l4d_field_desc* f /*host*/);
int64_t l4d_density_encode_fwd_workspace(const l4d_field_desc* f /*host*/, int64_t P);
int l4d_density_encode_fwd(const l4d_field_desc* f /*host*/, const float* xt, const void* flow16,
                           const float* tinfo, int64_t P, void* X, int32_t in_pad, void* hd_scratch, float* plane_rows,
                           void* stream);
/* l4d_density_encode_fwd followed by the density network's forward pass l4d_mlp_fwd_sigma(X, ...) (model/lidar4d.py:181-186) as ONE
 * call: sigma_weights = the network's fp16 weights (W1 [64, in_pad] | hidden [64, 64] x (n_hidden - 1) | Wo [16, 64]), y [P, 16] fp16,
 * act [n_hidden, P, 64] fp16 or null, sigma [P] fp32 = exp(y[:, 0]).  For the default shape (in_pad 128, one hidden layer) behind the
 * level-major encode the network runs as the encode kernel's epilogue on the rows it holds in LDS (bit-identical outputs; X is still
 * written); every other shape runs the two launches. */
int l4d_density_encode_sigma_fwd(const l4d_field_desc* f /*host*/, const float* xt, const void* flow16, const float* tinfo,
                                 int64_t P, void* X, int32_t in_pad, void* hd_scratch, float* plane_rows,
                                 const void* sigma_weights, int32_t n_hidden, void* y, void* act, float* sigma, void* stream);
/* Adjoint.  dX [P,in_pad] fp16 (loss-scaled); parameter gradients are accumulated multiplied by param_scale
 * (= 1/loss_scale); dflow16 [P,16] fp16 stays in dX's scaled domain.  plane_abs_max: device fp32 = max |plane
 * parameter| (bounds the fixed-point LDS accumulators); samples_per_ray: T when the P rows are rays x T samples in
 * ray-major order (enables skipping whole wavefronts per plane band), 0 if unknown; workspace: l4d_density_encode_bwd_workspace() bytes of
 * device scratch.  Several launches: see lidar4d_amd/csrc/field_bwd.hip.  With side streams enabled (l4d_streams_config bit 1)
 * the independent parts run concurrently; dflow16 is always produced on `stream`.  defer_join = 1: return without joining the
 * side streams -- the caller may queue the consumers of dflow16 on `stream` at once and must call l4d_streams_join(stream)
 * before reading any parameter gradient or releasing `workspace`. */
int64_t l4d_density_encode_bwd_workspace(const l4d_field_desc* f /*host*/, int64_t P);
int l4d_density_encode_bwd(const l4d_field_desc* f /*host*/, const l4d_field_grads* g /*host*/, const float* xt,
                           const void* flow16, const float* tinfo, int64_t P, const void* dX, int32_t in_pad,
                           float param_scale, const float* plane_abs_max, int32_t samples_per_ray, void* workspace,
                           void* dflow16, float* plane_rows /*null, or l4d_plane_rows_workspace() bytes: see the forward*/,
                           const float* gd_absmax /*null, or device fp32: max |dX[:, n_scales*C : 2*n_scales*C]| as l4d_mlp_bwd
                           reports it (dx_absmax): one pass over dX less -- the time-plane kernel then does the preparation pass' work*/,
                           int32_t defer_join, void* stream);

/* ---- chamfer_3DDist : utils/chamfer3D/chamfer3D.cu:11-194, dist_chamfer_3D.py:31-83 (SURVEY 8f "next" row 1) ----
 * xyz1 [b,n,3], xyz2 [b,m,3] fp32 -> dist1 [b,n], dist2 [b,m] (squared distance to the nearest point of the other
 * cloud), idx1 [b,n], idx2 [b,m] (int32, first minimum wins).  workspace: l4d_chamfer_workspace() bytes. */
int64_t l4d_chamfer_workspace(int32_t b, int32_t n, int32_t m);
int l4d_chamfer_fwd(const float* xyz1, const float* xyz2, int32_t b, int32_t n, int32_t m, float* dist1, float* dist2,
                    int32_t* idx1, int32_t* idx2, void* workspace, void* stream);
/* grad_xyz1 [b,n,3], grad_xyz2 [b,m,3] must be zero-filled by the caller (accumulated with atomics) */
int l4d_chamfer_bwd(const float* xyz1, const float* xyz2, int32_t b, int32_t n, int32_t m, const float* grad_dist1,
                    const float* grad_dist2, const int32_t* idx1, const int32_t* idx2, float* grad_xyz1,
                    float* grad_xyz2, void* stream);

/* ---- step glue: batch assembly and loss evaluation either side of the render path, one launch each (csrc/glue.hip) ----
 * l4d_lidar_ray_batch: rows / cols [n] int64 = the drawn pixels (data/base_dataset.py:36-70; cols are taken modulo W);
 * pose [4,4] row-major sensor-to-world; (fov_up, fov) in degrees; image [H,W,3] ground-truth range image or null.
 * -> rays_o, rays_d [n,3] (base_dataset.py:72-102), gt [n,3] = image at the pixels (kitti360_dataset.py:181-187),
 * inds [n] = row * W + col. */
int l4d_lidar_ray_batch(const int64_t* rows, const int64_t* cols, int32_t n, const float* pose, float fov_up, float fov,
                        int32_t H, int32_t W, const float* image, float* rays_o, float* rays_d, float* gt, int64_t* inds,
                        void* stream);
/* model/runner.py:179-213 with the default criteria: loss[0] = sum over rays of alpha_d |d - d_gt| m + alpha_r (r - clamp(m,
 * smooth, 1 - smooth))^2 + alpha_i ((i - i_gt) m)^2 with m = gt[:,0]; g_depth [n], g_image [n,2] = its gradients;
 * pts (or null) [2,n,3] = predicted / ground-truth points along the rays in metres (runner.py:215-218: rays_d * depth * m / scale). */
int64_t l4d_glue_workspace(int32_t n); /* floats of scratch (`partial`) the two calls below need for n rays */
int l4d_lidar_losses(const float* depth, const float* image, const float* gt, const float* rays_d, int32_t n, float alpha_d,
                     float alpha_r, float alpha_i, float smooth, float scale, float* loss, float* g_depth, float* g_image,
                     float* pts, float* partial, void* stream);
/* ray-chamfer term (runner.py:215-220) behind l4d_chamfer_fwd on pts: loss[0] += coef * sum(dist1 + dist2) and g_depth +=
 * its gradient wrt the rendered depth (coef = 0.5 / n / world for the reference's mean * 0.5). */
int l4d_ray_chamfer_grad(const float* pts, const float* rays_d, const float* gt, const float* dist1, const float* dist2,
                         const int32_t* idx1, const int32_t* idx2, int32_t n, float coef, float scale, float* loss,
                         float* g_depth, float* partial, void* stream);
/* out_a[i] = a[i] * s[0] (na values), out_b[i] = b[i] * s[0] (nb values); s on the device */
int l4d_scale_buffers(const float* a, float* out_a, int64_t na, const float* b, float* out_b, int64_t nb, const float* s,
                      void* stream);

/* scene-flow consistency loss (runner.py:222-253) around the flow field's own kernels (l4d_hashgrid_t_fwd / l4d_mlp_fwd and their
 * adjoints):
 *   l4d_flow_xt            xt [n,4] = [(pc + bound) / (2 bound), t[0]]                                   (lidar4d.py:133-137)
 *   l4d_flow_warp          out [v,n,3] = pc + float(y16[:, col0[v] .. +2]) * step[v], v < n_variants <= 4 (runner.py:233-247)
 *   l4d_flow_chamfer_grad  one chamfer term behind l4d_chamfer_fwd(p, q): partial[block] = share of sum(dist1) + sum(dist2);
 *                          dy [n,6] (zero-filled by the caller) += step * d(0.5 (sum dist1 + sum dist2)) / dp in columns col0 .. +2
 *   l4d_flow_loss_finish   loss[0] = 0.5 sum(partial) + w_ground sum |y_ground[:, :6]|; dy_g [ng,6] = w_ground sign(y_ground);
 *                          amax[0] = max |dy|, amax[1] = max |dy_g|
 *   l4d_flow_dy16          dy16 [n,16] fp16 = dy * g[0] * 2^k, k chosen on the device so that the largest entry lands in
 *                          [2^11, 2^12); inv_out[0] = 2^-k                                               (flow_field.py _FlowFn.backward)
 *   l4d_axpy_dev           y += a[0] * x, a on the device */
int l4d_flow_xt(const float* pc, int32_t n, const float* t, float bound, float* xt, void* stream);
int l4d_flow_warp(const float* pc, const void* y16, int32_t n, int32_t n_variants, const int32_t* col0, const float* step,
                  float* out, void* stream);
int l4d_flow_chamfer_grad(const float* p, int32_t n, const float* q, int32_t m, const float* dist1, const float* dist2,
                          const int32_t* idx1, const int32_t* idx2, float step, int32_t col0, float* dy, float* partial,
                          void* stream);
int l4d_flow_loss_finish(const float* partial, int32_t n_partial, const void* y_ground16, int32_t ng, float w_ground, float* dy_g,
                         const float* dy, int64_t n_dy, float* loss, float* amax, void* stream);
int l4d_flow_dy16(const float* dy, int32_t n, const float* g, const float* amax, void* dy16, float* inv_out, void* stream);
int l4d_axpy_dev(float* y, const float* x, int64_t n, const float* a, void* stream);

/* ---- range image <-> point cloud : utils/convert.py:4-156 (SURVEY 8f "next" row 4) ----
 * pano_to_lidar_with_intensities (convert.py:99-137): pano [H,W] fp32 range image (0 = no return), intensities [H,W]
 * or null (-> 0); lidar_K = (fov_up, fov) in degrees.  points: room for H*W rows [x,y,z,intensity]; the non-empty
 * pixels are written in row-major pixel order (np.where order), their number to *count (device int32).
 * workspace: l4d_pano_to_lidar_workspace() bytes. */
int64_t l4d_pano_to_lidar_workspace(int32_t H, int32_t W);
int l4d_pano_to_lidar(const float* pano, const float* intensities, int32_t H, int32_t W, double fov_up, double fov,
                      float* points, int32_t* count, void* workspace, void* stream);
/* lidar_to_pano_with_intensities (convert.py:4-66): points [n,4] fp32 (x,y,z,intensity) -> pano [H,W] = range of the
 * closest point that falls into each pixel (0 if none), intensities [H,W] (or null) = that point's intensity; of equally
 * close points the first in the array wins, like the reference's sequential loop.  Points at range >= max_depth or
 * outside the image are dropped.  workspace: l4d_lidar_to_pano_workspace() bytes. */
int64_t l4d_lidar_to_pano_workspace(int32_t H, int32_t W);
int l4d_lidar_to_pano(const float* points, int64_t n, int32_t H, int32_t W, double fov_up, double fov, float max_depth,
                      float* pano, float* intensities, void* workspace, void* stream);

/* ---- optimiser + casts (runner.py:506-508 Adam step; tcnn's per-forward fp32->fp16 param cast) ---- */
int l4d_cast_f32_to_f16(const float* src, void* dst, int64_t n, void* stream);
int l4d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_f16 /*or null*/,
                  int64_t n, float lr, float beta1, float beta2, float eps, float bias_c1, float bias_c2,
                  float grad_scale, void* stream);
/* The same update over n_ranges (<= 40) ranges [off[r], off[r] + len[r]) of the flat arenas in ONE launch, with the
 * optimiser's bookkeeping on the device so that the step never waits on the host (torch.optim.Adam keeps a step count per
 * parameter tensor and skips tensors whose .grad is None; torch.cuda.amp.GradScaler skips the whole step on a non-finite
 * gradient: runner.py:506-508).  off / len / lr / gate_idx: host arrays; off[r] a multiple of 4 elements.
 *   gates   device fp32 array or null; range r is updated only if gate_idx[r] < 0 or gates[gate_idx[r]] != 0
 *   scaler  device fp32[4] or null: [0] loss scale, [1] growth tracker, [2] != 0 -> skip every range, [3] 1 / loss scale
 *           (multiplied into the gradient together with grad_scale)
 *   steps   device int32[n_ranges]: per-range step counts, incremented here for the ranges that are updated; the bias
 *           corrections 1 - beta^t are derived from them on the device.
 *   sched   device fp32[2] or null: the reference's LambdaLR (main_lidar4d.py:303-305) on the device -- [0] iterations so
 *           far (incremented here on every call, skipped step or not), [1] this step's factor 0.1 ** min([0] / sched_iters, 1),
 *           multiplied into lr[r]; with it a captured step (hipGraph) has no argument that changes between replays. */
int l4d_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_f16 /*or null*/,
                         int32_t n_ranges, const int64_t* off /*host*/, const int64_t* len /*host*/, const float* lr /*host*/,
                         const int32_t* gate_idx /*host, or null*/, const float* gates, const float* scaler, int32_t* steps,
                         float beta1, float beta2, float eps, float grad_scale, float* sched, float sched_iters,
                         void* stream);
/* GradScaler pieces (torch.cuda.amp.GradScaler as the reference uses it, runner.py:102,506-508), device-resident state
 * scaler_state fp32[4] as above.  check: sets state[2] = 1 if any of grad[0..n) is inf / nan (call after the gradient
 * all-reduce, before the Adam step).  update: halve the scale after a non-finite step, multiply it by growth_factor after
 * growth_interval consecutive clean steps, clear state[2], refresh state[3] (call after the Adam step). */
int l4d_grad_nonfinite_check(const float* grad, int64_t n, float* scaler_state, void* stream);
int l4d_scaler_update(float* scaler_state, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream);
/* out[0] = max |x[i]| over n fp32 values (x 16-byte aligned), +inf if any of them is inf / nan.  Replaces the torch
 * reductions ``x.abs().max()`` in the step (lidar4d_amd/fused.py: bound of the plane values for the fixed-point accumulators;
 * flow_field.py: normalisation of the scene-flow adjoint): torch's multi-block reduce zeroes its semaphores with
 * hipMemsetAsync, and a captured step must not contain memset nodes (see chamfer.hip / DESIGN.md section 5). */
int l4d_absmax_f32(const float* x, int64_t n, float* out, void* stream);
/* gates[i1] = gates[i2] = 1 for the pair of time slices HashGridT.forward blends at time tinfo[0] (hash_field.py:79-85):
 * the slices whose tables receive a gradient in this step (tinfo: l4d_time_setup). */
int l4d_mark_time_slices(const float* tinfo, int32_t n_slices, float* gates, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIDAR4D_HIP_H */
