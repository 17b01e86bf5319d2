"""Tensor-level wrappers over the C ABI (one function per entry point of include/lidar4d_hip.h).

Every wrapper checks device / dtype / contiguity, allocates nothing it was not asked to, and launches
on the current torch stream.  Tensors must live on a HIP device: there is no CPU fallback.
"""
import ctypes as C
import os

import torch

from . import _lib


def call(name, *args):
    _lib.call(name, *args)


LEVEL_MAJOR_MIN_POINTS = 1 << 20          # stand-alone hash-grid forward: level-major through scratch from here on
PLANE_ROWS_MIN_POINTS = 1 << 12           # below this the row build (a launch) costs more than the saved taps
LDS_DYNHASH_MIN_POINTS = 1 << 15          # below this the per-sample direct-gather path wins (table fills dominate)
BINNED_SCATTER_MIN_RECORDS = 1 << 16  # below this the global-atomic path is cheaper than two extra launches
FLOW_LEVELS_MIN_POINTS = 1 << 18          # the flow grid level-major through scratch (l4d_hashgrid_t_fwd_ws) from here on


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype=None, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.HipExtensionError(
            f"{name} is on {t.device}: the lidar4d_amd hot path runs only on a HIP device (no CPU fallback; "
            "the CPU restatement is oracle/, test infrastructure)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def _i32s(vals):
    return (C.c_int32 * len(vals))(*vals)


def _i64s(vals):
    return (C.c_int64 * len(vals))(*vals)


def _ptrs(tensors):
    return (C.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


# ---- hash grid -----------------------------------------------------------------------------------
def hashgrid_fwd(meta, x, cols, table16, out=None, out_col=0, level_major=None):
    """x [P, S] fp32 (grid coords = columns ``cols``), table16 fp16 flat -> out [P, >= L*F] fp16."""
    _chk(x, torch.float32, "x"), _chk(table16, torch.float16, "table")
    P = x.shape[0]
    if out is None:
        out = torch.empty(P, meta.n_output_dims, dtype=torch.float16, device=x.device)
    _chk(out, torch.float16, "out")
    d = meta.desc()
    if level_major is None:
        level_major = P >= LEVEL_MAJOR_MIN_POINTS and max(meta.size) * meta.n_features * 2 >= (2 << 20)
    if level_major:  # one level at a time over the whole chip (every L2 holds that level's table), level-major scratch, then rows
        ws = torch.empty(_lib.lib().l4d_hashgrid_fwd_workspace(C.byref(d), P), dtype=torch.uint8, device=x.device)
        call("l4d_hashgrid_fwd_ws", C.byref(d), _p(x), P, x.stride(0), _i32s(list(cols)), _p(table16),
             C.c_void_p(out.data_ptr() + 2 * out_col), out.stride(0), _p(ws), _stream())
        return out
    call("l4d_hashgrid_fwd", C.byref(d), _p(x), P, x.stride(0), _i32s(list(cols)), _p(table16),
         C.c_void_p(out.data_ptr() + 2 * out_col), out.stride(0), _stream())
    return out


def hashgrid_bwd(meta, x, cols, dout, grad_table, grad_scale=1.0, dout_col=0):
    _chk(x, torch.float32, "x"), _chk(dout, None, "dout"), _chk(grad_table, torch.float32, "grad_table")
    is_half = dout.dtype == torch.float16
    if not is_half and dout.dtype != torch.float32:
        raise TypeError("dout must be fp16 or fp32")
    d = meta.desc()
    call("l4d_hashgrid_bwd", C.byref(d), _p(x), x.shape[0], x.stride(0), _i32s(list(cols)),
         C.c_void_p(dout.data_ptr() + dout.element_size() * dout_col), dout.stride(0), int(is_half), float(grad_scale),
         _p(grad_table), _stream())


def hashgrid_t_fwd(meta, x, cols, tables16, t_dev, out=None, out_col=0, half_out=False):
    """Fused time-blend + interpT.  tables16: list of fp16 tables (1 = no blend); t_dev: 1-element fp32 device tensor."""
    _chk(x, torch.float32, "x"), _chk(t_dev, torch.float32, "t")
    for tb in tables16:
        _chk(tb, torch.float16, "table")
    P = x.shape[0]
    width = meta.n_levels * meta.n_features // 4
    if out is None:
        out = torch.empty(P, width, dtype=torch.float16 if half_out else torch.float32, device=x.device)
    _chk(out, None, "out")
    d = meta.desc()
    if P >= FLOW_LEVELS_MIN_POINTS and meta.n_dims == 3 and meta.n_features == 8 and out.dtype == torch.float16:
        # the flow field's grid on a render batch: level-major through a scratch array (csrc/hashgrid.hip hashgrid_t_fwd_levels_kernel)
        ws = torch.empty(_lib.lib().l4d_hashgrid_t_fwd_workspace(C.byref(d), P), dtype=torch.uint8, device=x.device)
        call("l4d_hashgrid_t_fwd_ws", C.byref(d), _p(x), P, x.stride(0), _i32s(list(cols)), _ptrs(tables16), len(tables16),
             _p(t_dev), C.c_void_p(out.data_ptr() + out.element_size() * out_col), out.stride(0), 1, _p(ws), _stream())
        return out
    call("l4d_hashgrid_t_fwd", C.byref(d), _p(x), P, x.stride(0), _i32s(list(cols)), _ptrs(tables16), len(tables16),
         _p(t_dev), C.c_void_p(out.data_ptr() + out.element_size() * out_col), out.stride(0),
         int(out.dtype == torch.float16), _stream())
    return out


def hashgrid_t_bwd(meta, x, cols, n_slices, t_dev, dout, grad_tables, grad_scale=1.0, dout_col=0):
    """grad_tables: list of n_slices fp32 tensors or None (slices that *t does not select are never touched)."""
    _chk(x, torch.float32, "x"), _chk(dout, None, "dout")
    d = meta.desc()
    scratch = torch.empty(meta.n_entries * (meta.n_features // 4), dtype=torch.float32, device=x.device)
    ws = None
    if dout.dtype == torch.float16 and x.shape[0] * (1 << meta.n_dims) >= BINNED_SCATTER_MIN_RECORDS:
        ws = torch.empty(_lib.lib().l4d_hashgrid_t_bwd_workspace(C.byref(d), x.shape[0]), dtype=torch.uint8, device=x.device)
    call("l4d_hashgrid_t_bwd", C.byref(d), _p(x), x.shape[0], x.stride(0), _i32s(list(cols)), n_slices, _p(t_dev),
         C.c_void_p(dout.data_ptr() + dout.element_size() * dout_col), dout.stride(0), int(dout.dtype == torch.float16),
         float(grad_scale), _ptrs(grad_tables), _p(scratch), _p(ws), _stream())


# ---- planes ----------------------------------------------------------------------------------------
class PlaneLayout:
    """Channel-last arena geometry for Planes4D (comb order of itertools.combinations(range(4), 2))."""
    COMBS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))

    def __init__(self, resolutions, channels):
        self.res = [list(r) for r in resolutions]  # per scale (x, y, z, t)
        self.n_scales = len(self.res)
        self.C = channels
        self.off = []
        o = 0
        for r in self.res:
            for a, b in self.COMBS:
                self.off.append(o)
                o += r[a] * r[b] * channels
        self.numel = o
        self.res_flat = _i32s([v for r in self.res for v in r])
        self.off_flat = _i64s(self.off)

    def plane_shape(self, s, c):
        a, b = self.COMBS[c]
        return (1, self.C, self.res[s][b], self.res[s][a])


def planes_relayout(layout, planes, arena, to_channel_last=True, accumulate=False):
    """planes: list (scale-major, 6 per scale) of [1,C,H,W] fp32 tensors <-> arena (flat fp32 channel-last); one launch.
    accumulate (with to_channel_last=False): planes += arena (gradients straight into the parameters' .grad views)."""
    for pl in planes:
        _chk(pl, torch.float32, "plane")
    _chk(arena, torch.float32, "arena")
    mode = 1 if to_channel_last else (2 if accumulate else 0)
    call("l4d_planes_relayout", _ptrs(planes), layout.res_flat, layout.n_scales, layout.C, _p(arena), layout.off_flat, mode, _stream())


def planes_fwd(layout, arena, xt, which=0):
    _chk(arena, torch.float32, "arena"), _chk(xt, torch.float32, "xt")
    P = xt.shape[0]
    n_out = layout.n_scales * layout.C
    out_s = torch.empty(P, n_out, dtype=torch.float32, device=xt.device) if which != 2 else None
    out_d = torch.empty(P, n_out, dtype=torch.float32, device=xt.device) if which != 1 else None
    call("l4d_planes_fwd", _p(arena), layout.off_flat, layout.res_flat, layout.n_scales, layout.C, _p(xt), P, which,
         _p(out_s), _p(out_d), _stream())
    return out_s, out_d


def planes_bwd(layout, arena, xt, which, dout_s, dout_d, grad_arena, want_dxt):
    _chk(arena, torch.float32, "arena"), _chk(xt, torch.float32, "xt"), _chk(grad_arena, torch.float32, "grad_arena")
    _chk(dout_s, torch.float32, "dout_s"), _chk(dout_d, torch.float32, "dout_d")
    dxt = torch.empty_like(xt) if want_dxt else None
    call("l4d_planes_bwd", _p(arena), layout.off_flat, layout.res_flat, layout.n_scales, layout.C, _p(xt), xt.shape[0],
         which, _p(dout_s), _p(dout_d), _p(grad_arena), _p(dxt), _stream())
    return dxt


# ---- frequency encoding / MLP ----------------------------------------------------------------------
def freq_fwd(x, n_freq=12, out=None):
    _chk(x, torch.float32, "x")
    P, D = x.shape
    if out is None:
        out = torch.empty(P, D * n_freq * 2, dtype=torch.float16, device=x.device)
    call("l4d_freq_fwd", _p(x), P, D, n_freq, _p(out), out.stride(0), _stream())
    return out


def mlp_fwd(x16, weights16, n_hidden, save_act=True, n_rows=None, y=None, act=None):
    """x16 [P, in_pad] fp16 -> y [P,16] fp16 (+ act [n_hidden, P, 64] fp16)."""
    _chk(x16, torch.float16, "x"), _chk(weights16, torch.float16, "weights"), _chk(n_rows, torch.int32, "n_rows")
    P, in_pad = x16.shape
    if y is None:
        y = torch.empty(P, 16, dtype=torch.float16, device=x16.device)
    if save_act and act is None:
        act = torch.empty(n_hidden, P, 64, dtype=torch.float16, device=x16.device)
    call("l4d_mlp_fwd", _p(x16), P, _p(n_rows), in_pad, n_hidden, _p(weights16), _p(y), _p(act), _stream())
    return y, act


def mlp_recompute_supported(in_pad, n_hidden):
    """Shapes for which l4d_mlp_bwd recomputes the hidden activations itself (act = None), so that the forward does not store
    them: narrow inputs (the flow network: 32-byte rows against 256 B of activations; measured forward 1.11 -> 0.33 ms,
    backward 1.64 -> 1.41 ms at 12.6 M rows) and, since round 5, the 128 -> 64 -> 16 density network (round 2 had measured that
    trade neutral, forward 0.91 -> 0.67 ms, backward 1.93 -> 2.13 ms; with today's kernels it is -0.17 ms per step).  Wider / deeper
    shapes spill registers and keep their activations."""
    if in_pad == 128 and n_hidden == 1:
        # the density network (round 5): its forward runs as the encode kernel's epilogue and stores 128 B less per sample, the backward
        # reads 128 B less and recomputes the hidden layer from the row it reads anyway: 32.08 -> 31.91 ms per step (gpurun_out/s6;
        # L4D_MLP_RECOMP_SIGMA=0: store them, A/B)
        return os.environ.get("L4D_MLP_RECOMP_SIGMA") != "0"
    return in_pad <= 32 and 1 <= n_hidden <= 3


def mlp_bwd(x16, act, dy16, weights16, n_hidden, grad_w, inv_loss_scale, n_rows=None, want_dx=True, dx=None, dx_absmax=None,
            absmax_cols=(0, 0)):
    """dx_absmax: 1-element fp32 device tensor (zeroed by the caller) that receives max |dx[:, absmax_cols[0]:absmax_cols[1]]|
    (columns in multiples of 16; +inf if a value is not finite)."""
    _chk(x16, torch.float16, "x"), _chk(act, torch.float16, "act"), _chk(dy16, torch.float16, "dy")
    _chk(weights16, torch.float16, "weights"), _chk(grad_w, torch.float32, "grad_w"), _chk(n_rows, torch.int32, "n_rows")
    _chk(dx_absmax, torch.float32, "dx_absmax")
    P, in_pad = x16.shape
    if want_dx and dx is None:
        dx = torch.empty(P, in_pad, dtype=torch.float16, device=x16.device)
    call("l4d_mlp_bwd", _p(x16), _p(act), _p(dy16), P, _p(n_rows), in_pad, n_hidden, _p(weights16), _p(dx), _p(grad_w),
         float(inv_loss_scale), _p(dx_absmax), int(absmax_cols[0]), int(absmax_cols[1]), _stream())
    return dx


# ---- renderer --------------------------------------------------------------------------------------
def sample_rays(rays_o, rays_d, lin, noise, near, far, bound, want_xyz=True):
    _chk(rays_o, torch.float32, "rays_o"), _chk(rays_d, torch.float32, "rays_d"), _chk(lin, torch.float32, "lin")
    _chk(noise, torch.float32, "noise")
    N, T = rays_o.shape[0], lin.shape[0]
    z = torch.empty(N, T, dtype=torch.float32, device=rays_o.device)
    xyz = torch.empty(N * T, 3, dtype=torch.float32, device=rays_o.device) if want_xyz else None
    call("l4d_sample_rays", _p(rays_o), _p(rays_d), _p(lin), _p(noise), N, T, float(near), float(far), float(bound),
         _p(z), _p(xyz), _stream())
    return z, xyz


def sample_rays_xt(rays_o, rays_d, lin, noise, t_dev, near, far, bound):
    _chk(rays_o, torch.float32, "rays_o"), _chk(rays_d, torch.float32, "rays_d"), _chk(lin, torch.float32, "lin")
    _chk(noise, torch.float32, "noise"), _chk(t_dev, torch.float32, "t")
    N, T = rays_o.shape[0], lin.shape[0]
    z = torch.empty(N, T, dtype=torch.float32, device=rays_o.device)
    xt = torch.empty(N * T, 4, dtype=torch.float32, device=rays_o.device)
    call("l4d_sample_rays_xt", _p(rays_o), _p(rays_d), _p(lin), _p(noise), _p(t_dev), N, T, float(near), float(far),
         float(bound), _p(z), _p(xt), _stream())
    return z, xt


def composite_fwd(sigma, z_vals, sample_dist, density_scale, active_sensor, want_mask=True, want_idx=True):
    _chk(sigma, torch.float32, "sigma"), _chk(z_vals, torch.float32, "z_vals")
    N, T = z_vals.shape
    dev = z_vals.device
    weights = torch.empty(N, T, dtype=torch.float32, device=dev)
    wsum = torch.empty(N, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    mask = torch.empty(N, T, dtype=torch.uint8, device=dev) if want_mask else None
    idx = torch.empty(N * T, dtype=torch.int32, device=dev) if want_idx else None
    count = torch.empty(1, dtype=torch.int32, device=dev) if want_idx else None
    call("l4d_composite_fwd", _p(sigma), _p(z_vals), N, T, float(sample_dist), float(density_scale), int(active_sensor),
         _p(weights), _p(wsum), _p(depth), _p(mask), _p(idx), _p(count), _stream())
    return weights, wsum, depth, mask, idx, count


def composite_image(weights, attr, C_out):
    _chk(weights, torch.float32, "weights"), _chk(attr, torch.float32, "attr")
    N, T = weights.shape
    image = torch.empty(N, C_out, dtype=torch.float32, device=weights.device)
    call("l4d_composite_image", _p(weights), _p(attr), N, T, C_out, _p(image), _stream())
    return image


def composite_bwd(sigma, z_vals, weights, attr, C_out, sample_dist, density_scale, active_sensor, d_depth, d_wsum,
                  d_image, d_weights, want_d_attr=True):
    for nm, t in (("sigma", sigma), ("z_vals", z_vals), ("weights", weights), ("attr", attr), ("d_depth", d_depth),
                  ("d_wsum", d_wsum), ("d_image", d_image), ("d_weights", d_weights)):
        _chk(t, torch.float32, nm)
    N, T = z_vals.shape
    d_sigma = torch.empty(N, T, dtype=torch.float32, device=z_vals.device)
    d_attr = torch.empty(N * T, C_out, dtype=torch.float32, device=z_vals.device) if want_d_attr else None
    call("l4d_composite_bwd", _p(sigma), _p(z_vals), _p(weights), _p(attr), N, T, C_out, float(sample_dist),
         float(density_scale), int(active_sensor), _p(d_depth), _p(d_wsum), _p(d_image), _p(d_weights), _p(d_sigma),
         _p(d_attr), _stream())
    return d_sigma, d_attr


def attr_gather(idx, count, cap, T, dir_enc16, h16, n_geo, in_pad, xa=None):
    _chk(idx, torch.int32, "idx"), _chk(count, torch.int32, "count"), _chk(dir_enc16, torch.float16, "dir_enc")
    _chk(h16, torch.float16, "h")
    if xa is None:
        xa = torch.empty(cap, in_pad, dtype=torch.float16, device=h16.device)
    call("l4d_attr_gather", _p(idx), _p(count), cap, T, _p(dir_enc16), dir_enc16.shape[1], _p(h16), n_geo, _p(xa), in_pad,
         _stream())
    return xa


def attr_mlp_supported(in_pad, n_enc, n_geo):
    return in_pad == 96 and n_enc % 8 == 0 and n_enc // 16 == 4 and n_enc + 16 <= in_pad and n_geo == 15


def attr_mlp_fwd(idx, count, cap, T, dir_enc16, h16, n_geo, in_pad, weights16, n_hidden, save_act=True, x_rows_out=None,
                 attr_dense=None, attr_compact=None, channel=0):
    """Attribute network on the compacted work list, input rows assembled in the kernel -> y [cap,16] (+ act).
    x_rows_out: [cap, in_pad] fp16 that receives the assembled rows (physical column order) for attr_mlp_bwd.
    attr_dense [P,2] / attr_compact [cap,2] fp32: sigmoid epilogue into column ``channel`` (then y is not produced: None)."""
    _chk(idx, torch.int32, "idx"), _chk(count, torch.int32, "count"), _chk(dir_enc16, torch.float16, "dir_enc")
    _chk(h16, torch.float16, "h"), _chk(weights16, torch.float16, "weights"), _chk(x_rows_out, torch.float16, "x_rows_out")
    _chk(attr_dense, torch.float32, "attr_dense"), _chk(attr_compact, torch.float32, "attr_compact")
    y = torch.empty(cap, 16, dtype=torch.float16, device=h16.device) if attr_dense is None else None
    act = torch.empty(n_hidden, cap, 64, dtype=torch.float16, device=h16.device) if save_act else None
    call("l4d_attr_mlp_fwd", _p(idx), _p(count), cap, T, _p(dir_enc16), dir_enc16.shape[1], _p(h16), n_geo, in_pad, n_hidden,
         _p(weights16), _p(y), _p(act), _p(x_rows_out), _p(attr_dense), _p(attr_compact), int(channel), _stream())
    return y, act


def mlp_fwd_sigma(x16, weights16, n_hidden, save_act=True):
    """mlp_fwd + the density activation as epilogue -> (y [P,16] fp16, act, sigma [P] fp32 = exp(y[:, 0]))."""
    _chk(x16, torch.float16, "x"), _chk(weights16, torch.float16, "weights")
    P, in_pad = x16.shape
    y = torch.empty(P, 16, dtype=torch.float16, device=x16.device)
    act = torch.empty(n_hidden, P, 64, dtype=torch.float16, device=x16.device) if save_act else None
    sigma = torch.empty(P, dtype=torch.float32, device=x16.device)
    call("l4d_mlp_fwd_sigma", _p(x16), P, in_pad, n_hidden, _p(weights16), _p(y), _p(act), _p(sigma), _stream())
    return y, act, sigma


def attr_mlp_bwd(x_rows, count, n_enc, n_geo, act, dy16, weights16, n_hidden, grad_w, inv_loss_scale):
    """x_rows [cap, in_pad]: the rows attr_mlp_fwd stored.  -> dx_tail [cap, in_pad - 64] fp16: the input gradient's columns
    64 .. in_pad - 1, the 16 columns from n_enc - 64 on in the sigma network row's order [-, g0 .. g14]
    (attr_gather_bwd(..., h_layout=True))."""
    _chk(x_rows, torch.float16, "x_rows"), _chk(act, torch.float16, "act"), _chk(dy16, torch.float16, "dy")
    _chk(grad_w, torch.float32, "grad_w"), _chk(count, torch.int32, "count")
    cap, in_pad = x_rows.shape
    dx = torch.empty(cap, in_pad - 64, dtype=torch.float16, device=x_rows.device)
    call("l4d_attr_mlp_bwd", _p(x_rows), _p(count), cap, n_enc, n_geo, in_pad, n_hidden, _p(act), _p(dy16), _p(weights16), _p(dx),
         _p(grad_w), float(inv_loss_scale), _stream())
    return dx


def attr_mlp_bwd_gathered_supported(n_hidden):
    return n_hidden <= 2


def attr_mlp_recompute_supported(n_hidden):
    """Shapes for which l4d_attr_mlp_bwd_gathered recomputes the hidden activations (act = None): the forward then stores
    nothing but the sigmoid outputs."""
    return n_hidden <= 2


def attr_mlp_bwd_gathered(idx, count, cap, T, dir_enc16, h16, n_geo, in_pad, act, dy16, weights16, n_hidden, grad_w, inv_loss_scale,
                          d_attr=None, attr_compact=None, channel=0, loss_scale=1.0, dh16=None, accumulate=False):
    """attr_mlp_bwd with the rows assembled in the kernel again (nothing stored by the forward).
    dy16 given -> returns dx_tail [cap, in_pad - 64].  d_attr [P,2] / attr_compact [cap,2] / dh16 [P,16] given (dy16 None) ->
    the sigmoid-scatter adjoint is derived in the kernel and the geo-feature gradient is stored in / added to dh16; returns None."""
    _chk(idx, torch.int32, "idx"), _chk(count, torch.int32, "count"), _chk(dir_enc16, torch.float16, "dir_enc")
    _chk(h16, torch.float16, "h"), _chk(act, torch.float16, "act"), _chk(dy16, torch.float16, "dy")
    _chk(weights16, torch.float16, "weights"), _chk(grad_w, torch.float32, "grad_w")
    _chk(d_attr, torch.float32, "d_attr"), _chk(attr_compact, torch.float32, "attr_compact"), _chk(dh16, torch.float16, "dh")
    dx = torch.empty(cap, in_pad - 64, dtype=torch.float16, device=h16.device) if d_attr is None else None
    call("l4d_attr_mlp_bwd_gathered", _p(idx), _p(count), cap, T, _p(dir_enc16), dir_enc16.shape[1], _p(h16), n_geo, in_pad,
         n_hidden, _p(act), _p(dy16), _p(weights16), _p(dx), _p(grad_w), float(inv_loss_scale), _p(d_attr), _p(attr_compact),
         int(channel), float(loss_scale), _p(dh16), int(accumulate), _stream())
    return dx


def attr_scatter(idx, count, cap, y_raydrop, y_intensity, attr_dense, attr_compact):
    call("l4d_attr_scatter", _p(idx), _p(count), cap, _p(y_raydrop), _p(y_intensity), _p(attr_dense), _p(attr_compact),
         _stream())


def attr_scatter_bwd(idx, count, cap, d_attr, attr_compact, loss_scale, dy_r, dy_i):
    call("l4d_attr_scatter_bwd", _p(idx), _p(count), cap, _p(d_attr), _p(attr_compact), float(loss_scale), _p(dy_r),
         _p(dy_i), _stream())


def attr_gather_bwd(idx, count, cap, dxa_r, dxa_i, in_pad, n_enc, n_geo, dh16, h_layout=False):
    call("l4d_attr_gather_bwd", _p(idx), _p(count), cap, _p(dxa_r), _p(dxa_i), in_pad, n_enc, n_geo, _p(dh16), int(h_layout), _stream())


def sigma_from_h(h16):
    _chk(h16, torch.float16, "h")
    sigma = torch.empty(h16.shape[0], dtype=torch.float32, device=h16.device)
    call("l4d_sigma_from_h", _p(h16), h16.shape[0], _p(sigma), _stream())
    return sigma


def sigma_bwd(h16, d_sigma, loss_scale, dh16):
    _chk(h16, torch.float16, "h"), _chk(d_sigma, torch.float32, "d_sigma"), _chk(dh16, torch.float16, "dh")
    call("l4d_sigma_bwd", _p(h16), _p(d_sigma), h16.shape[0], float(loss_scale), _p(dh16), _stream())


def sigma_bwd_rows(sigma, d_sigma, loss_scale, dh16):
    """dh16 [P,16] := [d_sigma * exp(clamp(h0)) * loss_scale, 0 x 15] in one dense pass (no zero fill, no strided column access)."""
    _chk(sigma, torch.float32, "sigma"), _chk(d_sigma, torch.float32, "d_sigma"), _chk(dh16, torch.float16, "dh")
    call("l4d_sigma_bwd_rows", _p(sigma), _p(d_sigma), sigma.numel(), float(loss_scale), _p(dh16), _stream())


# ---- fused field -----------------------------------------------------------------------------------
def time_setup(t_dev, num_frames, tinfo=None):
    _chk(t_dev, torch.float32, "t")
    if tinfo is None:
        tinfo = torch.empty(8, dtype=torch.float32, device=t_dev.device)
    call("l4d_time_setup", _p(t_dev), num_frames, _p(tinfo), _stream())
    return tinfo


def density_encode_fwd(field_desc, xt, flow16, tinfo, in_pad, X=None, sigma_weights16=None, n_hidden=0, save_act=True):
    """-> X [P, in_pad] fp16.  With ``sigma_weights16`` (the density network's fp16 weights): -> (X, y [P,16] fp16, act or None,
    sigma [P] fp32) = the rows AND ``mlp_fwd_sigma`` of them in one call (l4d_density_encode_sigma_fwd: for the default network
    shape the network runs inside the encode kernel)."""
    _chk(xt, torch.float32, "xt"), _chk(flow16, torch.float16, "flow16"), _chk(tinfo, torch.float32, "tinfo")
    P = xt.shape[0]
    if X is None:
        X = torch.empty(P, in_pad, dtype=torch.float16, device=xt.device)
    scratch = None
    if P >= LDS_DYNHASH_MIN_POINTS and max(field_desc.hash_dynamic[1].size[field_desc.hash_dynamic[1].n_levels - 1],
                                           field_desc.hash_dynamic[2].size[field_desc.hash_dynamic[2].n_levels - 1]) <= 8192:
        scratch = torch.empty(_lib.lib().l4d_density_encode_fwd_workspace(C.byref(field_desc), P), dtype=torch.uint8, device=xt.device)
    rows = None
    if P >= PLANE_ROWS_MIN_POINTS:  # time planes through per-call 1-D rows (two taps instead of four)
        rows = torch.empty(_lib.lib().l4d_plane_rows_workspace(C.byref(field_desc)) // 4, dtype=torch.float32, device=xt.device)
    if sigma_weights16 is not None:
        _chk(sigma_weights16, torch.float16, "sigma weights")
        y = torch.empty(P, 16, dtype=torch.float16, device=xt.device)
        act = torch.empty(n_hidden, P, 64, dtype=torch.float16, device=xt.device) if save_act else None
        sigma = torch.empty(P, dtype=torch.float32, device=xt.device)
        call("l4d_density_encode_sigma_fwd", C.byref(field_desc), _p(xt), _p(flow16), _p(tinfo), P, _p(X), in_pad, _p(scratch), _p(rows),
             _p(sigma_weights16), int(n_hidden), _p(y), _p(act), _p(sigma), _stream())
        return X, y, act, sigma
    call("l4d_density_encode_fwd", C.byref(field_desc), _p(xt), _p(flow16), _p(tinfo), P, _p(X), in_pad, _p(scratch), _p(rows), _stream())
    return X


def density_encode_bwd(field_desc, field_grads, xt, flow16, tinfo, dX, param_scale, plane_abs_max, samples_per_ray=0,
                       dflow16=None, defer_join=False, gd_absmax=None):
    """Adjoint of density_encode_fwd (several launches, lidar4d_amd/csrc/field_bwd.hip).  plane_abs_max: 1-element fp32
    device tensor, max |plane parameter| (bound for the fixed-point LDS accumulators).
    defer_join: return (dflow16, keepalive) with the library's side streams still running; the caller queues the consumers of
    dflow16, then calls ``streams_join()`` and only then drops ``keepalive`` (the workspace the side streams write).
    gd_absmax: 1-element fp32 device tensor = max |dX[:, n_scales*C : 2*n_scales*C]| (``mlp_bwd(dx_absmax=...)``): the separate
    preparation pass over dX is then folded into the time-plane kernel."""
    _chk(dX, torch.float16, "dX"), _chk(plane_abs_max, torch.float32, "plane_abs_max")
    P, in_pad = dX.shape
    if dflow16 is None:
        dflow16 = torch.empty(P, 16, dtype=torch.float16, device=dX.device)
    nbytes = _lib.lib().l4d_density_encode_bwd_workspace(C.byref(field_desc), P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dX.device)
    rows = None
    if P >= PLANE_ROWS_MIN_POINTS:
        rows = torch.empty(_lib.lib().l4d_plane_rows_workspace(C.byref(field_desc)) // 4, dtype=torch.float32, device=dX.device)
    call("l4d_density_encode_bwd", C.byref(field_desc), C.byref(field_grads), _p(xt), _p(flow16), _p(tinfo), P, _p(dX),
         in_pad, float(param_scale), _p(plane_abs_max), int(samples_per_ray), _p(ws), _p(dflow16), _p(rows), _p(gd_absmax),
         int(bool(defer_join)), _stream())
    if defer_join:
        return dflow16, (ws, rows)
    return dflow16


def streams_join():
    """The current stream waits for everything outstanding on the library's side streams (l4d_streams_join)."""
    call("l4d_streams_join", _stream())


def side_fork(i):
    """Library side stream i (0..2) as a torch stream that continues from the current stream's end (l4d_side_fork)."""
    h = _lib.lib().l4d_side_fork(_stream(), int(i))
    if not h:
        raise _lib.HipExtensionError("l4d_side_fork failed: " + _lib.lib().l4d_last_error().decode())
    return torch.cuda.ExternalStream(h)


def side_join(i):
    call("l4d_side_join", _stream(), int(i))


def streams_mask():
    return int(_lib.lib().l4d_streams_mask())


# ---- optimiser / casts -------------------------------------------------------------------------------
def cast_f32_to_f16(src, dst=None):
    _chk(src, torch.float32, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float16, device=src.device)
    _chk(dst, torch.float16, "dst")
    call("l4d_cast_f32_to_f16", _p(src), _p(dst), src.numel(), _stream())
    return dst


def adam_step(param, grad, exp_avg, exp_avg_sq, param16, lr, beta1, beta2, eps, step, grad_scale=1.0):
    for nm, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _chk(t, torch.float32, nm)
    _chk(param16, torch.float16, "param16")
    call("l4d_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(param16), param.numel(), float(lr),
         float(beta1), float(beta2), float(eps), 1.0 - beta1 ** step, 1.0 - beta2 ** step, float(grad_scale), _stream())


class AdamRanges:
    """Host-side description of the ranges one l4d_adam_step_ranges launch walks (built once)."""

    def __init__(self, offs, lens, lr_mults, gate_idx):
        self.n = len(offs)
        self.offs, self.lens, self.lr_mults, self.gate_idx = list(offs), list(lens), list(lr_mults), list(gate_idx)
        self._off = _i64s(self.offs)
        self._len = _i64s(self.lens)
        self._gate = _i32s(self.gate_idx)


def adam_step_ranges(param, grad, exp_avg, exp_avg_sq, param16, ranges, lr, gates, scaler_state, steps, beta1, beta2, eps,
                     grad_scale=1.0, sched=None, sched_iters=1.0):
    """One launch over all ranges; per-range step counters / gates / the scaler's skip flag live on the device.
    sched: device fp32[2] = [iterations so far, this step's factor]: the learning-rate schedule 0.1 ** min(it / sched_iters, 1)
    evaluated on the device and multiplied into ``lr`` (pass the base rate then); None: ``lr`` is used as given."""
    for nm, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _chk(t, torch.float32, nm)
    _chk(param16, torch.float16, "param16"), _chk(gates, torch.float32, "gates"), _chk(scaler_state, torch.float32, "scaler")
    _chk(steps, torch.int32, "steps")
    lrs = (C.c_float * ranges.n)(*[lr * m for m in ranges.lr_mults])
    call("l4d_adam_step_ranges", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(param16), ranges.n, ranges._off,
         ranges._len, C.cast(lrs, C.c_void_p), ranges._gate, _p(gates), _p(scaler_state), _p(steps), float(beta1), float(beta2),
         float(eps), float(grad_scale), _p(sched), float(sched_iters), _stream())


def grad_nonfinite_check(grad, scaler_state):
    _chk(grad, torch.float32, "grad"), _chk(scaler_state, torch.float32, "scaler")
    call("l4d_grad_nonfinite_check", _p(grad), grad.numel(), _p(scaler_state), _stream())


def absmax(x):
    """1-element fp32 device tensor max |x| (+inf if x holds inf / nan) by one HIP launch (l4d_absmax_f32) -- not torch's
    ``x.abs().max()``: its multi-block reduce zeroes a semaphore buffer with hipMemsetAsync, and memset nodes make a
    captured training step misbehave from its second replay on."""
    _chk(x, torch.float32, "x")
    x = x.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    call("l4d_absmax_f32", _p(x), x.numel(), _p(out), _stream())
    return out


def scaler_update(scaler_state, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    _chk(scaler_state, torch.float32, "scaler")
    call("l4d_scaler_update", _p(scaler_state), float(growth_factor), float(backoff_factor), int(growth_interval), _stream())


def mark_time_slices(tinfo, n_slices, gates):
    _chk(tinfo, torch.float32, "tinfo"), _chk(gates, torch.float32, "gates")
    call("l4d_mark_time_slices", _p(tinfo), int(n_slices), _p(gates), _stream())


def dyn_pairs_build(slice_tables16, pairs):
    """pairs [n_slices - 1, n_entries, 8] fp16 <- pair-interleaved copy of one plane's time-slice tables."""
    for tb in slice_tables16:
        _chk(tb, torch.float16, "slice table")
    _chk(pairs, torch.float16, "pairs")
    n_entries = slice_tables16[0].numel() // 4
    call("l4d_dyn_pairs_build", _ptrs(slice_tables16), len(slice_tables16), n_entries, _p(pairs), _stream())


# ---- step glue (csrc/glue.hip): batch assembly and loss evaluation, one launch each ------------------------------------------
def lidar_ray_batch(rows, cols, pose, fov, H, W, image=None):
    """Drawn pixels (rows / cols [n] int64 on the device) of one frame -> rays_o, rays_d [1,n,3], gt [1,n,3] (None without
    ``image`` [H,W,3]), inds [1,n]: data/base_dataset.py:72-102 and the gather of kitti360_dataset.py:181-187 in one launch."""
    _chk(rows, torch.int64, "rows"), _chk(cols, torch.int64, "cols"), _chk(pose, torch.float32, "pose")
    n, dev = rows.numel(), rows.device
    pose = pose.reshape(4, 4).contiguous()
    rays_o = torch.empty(1, n, 3, dtype=torch.float32, device=dev)
    rays_d = torch.empty(1, n, 3, dtype=torch.float32, device=dev)
    inds = torch.empty(1, n, dtype=torch.int64, device=dev)
    gt = None
    if image is not None:
        _chk(image, torch.float32, "image")
        image = image.contiguous()
        gt = torch.empty(1, n, 3, dtype=torch.float32, device=dev)
    call("l4d_lidar_ray_batch", _p(rows.contiguous()), _p(cols.contiguous()), n, _p(pose), float(fov[0]), float(fov[1]), int(H), int(W),
         _p(image), _p(rays_o), _p(rays_d), _p(gt), _p(inds), _stream())
    return rays_o, rays_d, gt, inds


def lidar_losses(depth, image, gt, rays_d, alpha_d, alpha_r, alpha_i, smooth, scale, want_points):
    """runner.py:179-213 (default criteria) -> loss [1], g_depth [n], g_image [n,2], pts [2,n,3] or None."""
    for t, nm in ((depth, "depth"), (image, "image"), (gt, "gt"), (rays_d, "rays_d")):
        _chk(t, torch.float32, nm)
    n, dev = depth.numel(), depth.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    g_depth = torch.empty(n, dtype=torch.float32, device=dev)
    g_image = torch.empty(n, 2, dtype=torch.float32, device=dev)
    pts = torch.empty(2, n, 3, dtype=torch.float32, device=dev) if want_points else None
    partial = torch.empty((n + 255) // 256 + 1, dtype=torch.float32, device=dev)
    call("l4d_lidar_losses", _p(depth), _p(image), _p(gt), _p(rays_d), n, float(alpha_d), float(alpha_r), float(alpha_i), float(smooth),
         float(scale), _p(loss), _p(g_depth), _p(g_image), _p(pts), _p(partial), _stream())
    return loss, g_depth, g_image, pts


def ray_chamfer_accumulate(pts, rays_d, gt, coef, scale, loss, g_depth):
    """Chamfer distance between pts[0] and pts[1] (l4d_chamfer_fwd), then loss += coef * sum(dist1 + dist2) and g_depth += its
    gradient wrt the rendered depth (l4d_ray_chamfer_grad): runner.py:215-220 with its autograd, three + five launches."""
    n, dev = pts.shape[1], pts.device
    if n == 0:
        return
    dist = torch.empty(2, n, dtype=torch.float32, device=dev)
    idx = torch.empty(2, n, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.lib().l4d_chamfer_workspace(1, n, n), dtype=torch.uint8, device=dev)
    call("l4d_chamfer_fwd", _p(pts[0]), _p(pts[1]), 1, n, n, _p(dist[0]), _p(dist[1]), _p(idx[0]), _p(idx[1]), _p(ws), _stream())
    partial = torch.empty((n + 255) // 256 + 1, dtype=torch.float32, device=dev)
    call("l4d_ray_chamfer_grad", _p(pts), _p(rays_d), _p(gt), _p(dist[0]), _p(dist[1]), _p(idx[0]), _p(idx[1]), n, float(coef), float(scale),
         _p(loss), _p(g_depth), _p(partial), _stream())


def scale_buffers(a, b, s):
    """(a * s[0], b * s[0]) in one launch; s a one-element fp32 tensor on the device."""
    out_a, out_b = torch.empty_like(a), torch.empty_like(b)
    call("l4d_scale_buffers", _p(a), _p(out_a), a.numel(), _p(b), _p(out_b), b.numel(), _p(s), _stream())
    return out_a, out_b


def flow_xt(pc, t_dev, bound):
    """[(pc + bound) / (2 bound), t] as [n,4] fp32 (lidar4d.py:133-137)."""
    _chk(pc, torch.float32, "pc"), _chk(t_dev, torch.float32, "t")
    n = pc.shape[0]
    xt = torch.empty(n, 4, dtype=torch.float32, device=pc.device)
    call("l4d_flow_xt", _p(pc), n, _p(t_dev), float(bound), _p(xt), _stream())
    return xt


def flow_warp(pc, y16, variants):
    """variants: [(col0, step)] (<= 4) -> [v, n, 3] fp32: pc + float(y16[:, col0:col0+3]) * step."""
    _chk(pc, torch.float32, "pc"), _chk(y16, torch.float16, "y16")
    n, nv = pc.shape[0], len(variants)
    out = torch.empty(nv, n, 3, dtype=torch.float32, device=pc.device)
    col0 = (C.c_int32 * 4)(*[int(c) for c, _ in variants])
    step = (C.c_float * 4)(*[float(s) for _, s in variants])
    call("l4d_flow_warp", _p(pc), _p(y16), n, nv, col0, C.cast(step, C.c_void_p), _p(out), _stream())
    return out
