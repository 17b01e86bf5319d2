"""Fused HIP render pipeline behind ``LiDAR4D.run`` (reference call stack SURVEY.md 3.1:
renderer.py:44-140 -> lidar4d.py:139-223).

Forward, 14 launches, no host synchronisation (the reference needs 4: lidar4d.py:143,201,203, hash_field.py:82):

  time_setup -> sample_rays_xt -> flow grid+interpT -> flow MLP -> fused field encode -> sigma MLP -> trunc_exp
  -> composite (weights, depth, wave-level mask compaction) -> frequency encode (per ray) -> attribute gather
  -> raydrop MLP, intensity MLP (device-sized work list) -> sigmoid+scatter -> image

Backward is the exact adjoint, accumulating parameter gradients straight into the model's flat gradient arena
(ParamStore); autograd sees one node.  fp16 adjoint operands carry ``model.loss_scale`` (tiny-cuda-nn uses the
same device: SURVEY A.1/A.3, loss_scale 128).
"""
import numpy as np
import torch

from . import ops
from ._lib import FieldDesc, FieldGrads


def _field_desc(model):
    store = model._store
    f16 = store.refresh16()
    arena = model.planes_encoder._arena()
    key = (f16.data_ptr(), arena.data_ptr())
    if getattr(model, "_fd_key", None) == key:
        _refresh_pairs(model)
        return model._fd
    he, pe = model.hash_encoder, model.planes_encoder
    fd = FieldDesc()
    fd.hash_static = he.hash_static.meta.desc()
    fd.hash_static_table = store.half(he.hash_static.params).data_ptr()
    fd.n_slices = he.hash_dynamic[0].time_resolution
    for p in range(3):
        fd.hash_dynamic[p] = he.hash_dynamic[p].meta.desc()
        for s in range(fd.n_slices):
            fd.hash_dynamic_tables[p][s] = store.half(he.hash_dynamic[p].hash_t[s].params).data_ptr()
    lay = pe.layout
    fd.n_scales, fd.plane_channels = lay.n_scales, lay.C
    for i, v in enumerate([v for r in lay.res for v in r]):
        fd.plane_res[i] = v
    for i, v in enumerate(lay.off):
        fd.plane_off[i] = v
    fd.planes_cl = arena.data_ptr()
    # pair-interleaved copies of the time-slice tables (both slices of a corner in one 16-byte load)
    model._dyn_pairs, model._pairs_version = [], None
    if fd.n_slices >= 2:
        for p in range(3):
            n_entries = he.hash_dynamic[p].hash_t[0].params.numel() // he.hash_dynamic[p].hash_t[0].meta.n_features
            buf = torch.empty(fd.n_slices - 1, n_entries, 8, dtype=torch.float16, device=f16.device)
            model._dyn_pairs.append(buf)
            fd.hash_dynamic_pairs[p] = buf.data_ptr()
    model._fd, model._fd_key = fd, key
    _refresh_pairs(model)
    return fd


def _refresh_pairs(model):
    store = model._store
    if not model._dyn_pairs or model._pairs_version == store.version16:
        return
    for p, buf in enumerate(model._dyn_pairs):
        hd = model.hash_encoder.hash_dynamic[p]
        ops.dyn_pairs_build([store.half(enc.params) for enc in hd.hash_t], buf)
    model._pairs_version = store.version16


def _field_grads(model, planes_grad_cl):
    store = model._store
    he = model.hash_encoder
    fg = FieldGrads()
    fg.hash_static_table = store.grad_view(he.hash_static.params).data_ptr()
    for p in range(3):
        for s in range(he.hash_dynamic[p].time_resolution):
            fg.hash_dynamic_tables[p][s] = store.grad_view(he.hash_dynamic[p].hash_t[s].params).data_ptr()
    fg.planes_cl = planes_grad_cl.data_ptr()
    return fg


def _flow_w16(model):
    store = model._store
    w0 = model.flow_net.linears()[0].weight
    off, _ = store.by_param[id(w0)]
    return store.refresh16()[off:off + model.flow_net.weight_numel16()]


def _flow_wgrad(model):
    store = model._store
    w0 = model.flow_net.linears()[0].weight
    off, _ = store.by_param[id(w0)]
    return store.flat_grad[off:off + model.flow_net.weight_numel16()]


class RenderFn(torch.autograd.Function):
    # The reference calls render() inside torch.cuda.amp.autocast (runner.py:497).  The node fixes its own precisions
    # (fp16 tables / MFMA operands, fp32 accumulation and compositing), so autocast is switched off inside it and
    # floating-point inputs arrive as fp32.
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, model, rays_o, rays_d, t_dev, noise, num_steps, train, *params):
        store = model._store
        N, T = rays_o.shape[0], int(num_steps)
        P = N * T
        dev = rays_o.device
        near32, far32 = np.float32(model.near_lidar), np.float32(model.far_lidar)
        sample_dist = float(np.float32(far32 - near32) / np.float32(T))
        lin = model._lin(T, dev)

        tinfo = ops.time_setup(t_dev, model.num_frames)
        z_vals, xt = ops.sample_rays_xt(rays_o, rays_d, lin, noise, t_dev, float(near32), float(far32), model.bound)

        # flow field (flow_field.py:113-130)
        fn = model.flow_net
        xf = ops.hashgrid_t_fwd(fn.grid_enc.meta, xt, (0, 1, 2), [store.half(fn.grid_enc.params)], t_dev, half_out=True)
        # hidden activations are only stored where the backward cannot recompute them from the input rows (ops.mlp_recompute_supported)
        flow16, act_f = ops.mlp_fwd(xf, _flow_w16(model), fn.n_hidden, save_act=train and not ops.mlp_recompute_supported(xf.shape[1], fn.n_hidden))

        # density (lidar4d.py:139-188)
        fd = _field_desc(model)
        # ... and the density network with its activation (trunc_exp, activation.py:6-20) in the same call: for the default shape it
        # runs as the encode kernel's epilogue on the rows in LDS (csrc/fused.hip SIGMA)
        sn = model.sigma_net
        X, h, act_s, sigma = ops.density_encode_fwd(fd, xt, flow16, tinfo, sn.in_pad, sigma_weights16=store.half(sn.params), n_hidden=sn.n_hidden_layers,
                                                    save_act=train and not ops.mlp_recompute_supported(sn.in_pad, sn.n_hidden_layers))

        # compositing + mask compaction (renderer.py:98-110)
        weights, wsum, depth, _, idx, count = ops.composite_fwd(sigma, z_vals, sample_dist, model.density_scale,
                                                                model.active_sensor, want_mask=False, want_idx=True)

        # attribute (lidar4d.py:191-223) on the compacted work list
        denc = ops.freq_fwd(((rays_d + 1) / 2).contiguous(), model.view_encoder.n_frequencies)
        an = model.intensity_net
        gathered = ops.attr_mlp_supported(an.in_pad, denc.shape[1], model.geo_feat_dim)
        attr = torch.zeros(P, 2, dtype=torch.float32, device=dev)
        attr_c = torch.empty(P, 2, dtype=torch.float32, device=dev)
        if gathered:  # the networks assemble their input rows themselves, forward and backward (three hidden layers: the first
            # one stores them for the backward pass), and apply the sigmoid + scatter into the dense [P, 2] image as their
            # epilogue (lidar4d.py:210-219)
            keep_rows = train and not ops.attr_mlp_bwd_gathered_supported(an.n_hidden_layers)
            XA = torch.empty(P, an.in_pad, dtype=torch.float16, device=dev) if keep_rows else None
            keep_act = train and (keep_rows or not ops.attr_mlp_recompute_supported(an.n_hidden_layers))  # else recomputed in the backward
            _, actR = ops.attr_mlp_fwd(idx, count, P, T, denc, h, model.geo_feat_dim, an.in_pad, store.half(model.raydrop_net.params),
                                       an.n_hidden_layers, save_act=keep_act, x_rows_out=XA, attr_dense=attr, attr_compact=attr_c, channel=0)
            _, actI = ops.attr_mlp_fwd(idx, count, P, T, denc, h, model.geo_feat_dim, an.in_pad, store.half(model.intensity_net.params),
                                       an.n_hidden_layers, save_act=keep_act, attr_dense=attr, attr_compact=attr_c, channel=1)
        else:
            XA = ops.attr_gather(idx, count, P, T, denc, h, model.geo_feat_dim, an.in_pad)
            yR, actR = ops.mlp_fwd(XA, store.half(model.raydrop_net.params), an.n_hidden_layers, save_act=train, n_rows=count)
            yI, actI = ops.mlp_fwd(XA, store.half(model.intensity_net.params), an.n_hidden_layers, save_act=train, n_rows=count)
            ops.attr_scatter(idx, count, P, yR, yI, attr, attr_c)
        image = ops.composite_image(weights, attr, 2)

        if train:
            ctx.model, ctx.T, ctx.sample_dist, ctx.gathered = model, T, sample_dist, gathered
            # the slice pair of THIS forward (LiDAR4D.run sets it per call; a later forward -- gradient accumulation, a no-grad
            # render -- must not change what this node's backward sees)
            ctx.slice_pair = getattr(model, "_host_slice_pair", None)
            ctx.save_for_backward(t_dev, tinfo, z_vals, xt, xf, flow16, act_f, X, h, act_s, sigma, weights, idx, count,
                                  XA, actR, actI, attr, attr_c, denc)
        ctx.mark_non_differentiable(z_vals, idx, count)
        return depth, image, wsum, weights, z_vals, idx, count

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_depth, d_image, d_wsum, d_weights, _dz, _di, _dc):
        model, T, sample_dist = ctx.model, ctx.T, ctx.sample_dist
        (t_dev, tinfo, z_vals, xt, xf, flow16, act_f, X, h, act_s, sigma, weights, idx, count, XA, actR, actI, attr,
         attr_c, denc) = ctx.saved_tensors
        store = model._store
        store.prepare_grads()
        # the current frame's time-slice pair is what receives dynamic-hash gradients (hash_field.py:79-85): raise its
        # gates for the optimiser (trainer.FlatAdam leaves ungated slices alone, like torch.optim.Adam with grad = None)
        ops.mark_time_slices(tinfo, model.hash_encoder.hash_dynamic[0].time_resolution, store.gates)
        ls = float(model.loss_scale)
        inv = 1.0 / ls
        P = xt.shape[0]
        dev = xt.device
        c = lambda t: None if t is None else t.float().contiguous()

        d_sigma, d_attr = ops.composite_bwd(sigma, z_vals, weights, attr, 2, sample_dist, model.density_scale,
                                            model.active_sensor, c(d_depth), c(d_wsum), c(d_image), c(d_weights))
        # attribute networks
        an = model.intensity_net
        n_enc = model.view_encoder.n_output_dims
        sigma_done = False
        if ctx.gathered and XA is None:
            # rows assembled again; the sigmoid-scatter adjoint in front of each network and the sum + scatter of the two
            # geo-feature gradients behind them run inside the kernels (first network stores into dh, second adds).  dh starts as
            # whole rows [density activation's adjoint, 0 x 15] (one dense pass instead of a zero fill + a strided column pass
            # afterwards); the networks keep that column (accumulate bit 1)
            dh = torch.empty(P, 16, dtype=torch.float16, device=dev)
            ops.sigma_bwd_rows(sigma.view(-1), d_sigma.view(-1), ls, dh)
            sigma_done = True
            for ch, (net, a_) in enumerate(((model.raydrop_net, actR), (model.intensity_net, actI))):
                ops.attr_mlp_bwd_gathered(idx, count, P, T, denc, h, model.geo_feat_dim, an.in_pad, a_, None, store.half(net.params),
                                          an.n_hidden_layers, store.grad_view(net.params), inv, d_attr=d_attr, attr_compact=attr_c,
                                          channel=ch, loss_scale=ls, dh16=dh, accumulate=(1 if ch == 1 else 0) | 2)
        else:
            dh = torch.zeros(P, 16, dtype=torch.float16, device=dev)
            dyR = torch.empty(P, 16, dtype=torch.float16, device=dev)
            dyI = torch.empty(P, 16, dtype=torch.float16, device=dev)
            ops.attr_scatter_bwd(idx, count, P, d_attr, attr_c, ls, dyR, dyI)
            if ctx.gathered:  # rows in the forward's physical column order (three hidden layers)
                dxaR = ops.attr_mlp_bwd(XA, count, n_enc, model.geo_feat_dim, actR, dyR, store.half(model.raydrop_net.params),
                                        an.n_hidden_layers, store.grad_view(model.raydrop_net.params), inv)
                dxaI = ops.attr_mlp_bwd(XA, count, n_enc, model.geo_feat_dim, actI, dyI, store.half(model.intensity_net.params),
                                        an.n_hidden_layers, store.grad_view(model.intensity_net.params), inv)
                ops.attr_gather_bwd(idx, count, P, dxaR, dxaI, an.in_pad - 64, n_enc - 64, model.geo_feat_dim, dh, h_layout=True)
            else:
                dxaR = ops.mlp_bwd(XA, actR, dyR, store.half(model.raydrop_net.params), an.n_hidden_layers,
                                   store.grad_view(model.raydrop_net.params), inv, n_rows=count)
                dxaI = ops.mlp_bwd(XA, actI, dyI, store.half(model.intensity_net.params), an.n_hidden_layers,
                                   store.grad_view(model.intensity_net.params), inv, n_rows=count)
                ops.attr_gather_bwd(idx, count, P, dxaR, dxaI, an.in_pad, n_enc, model.geo_feat_dim, dh)
        if not sigma_done:
            ops.sigma_bwd(h, d_sigma.view(-1), ls, dh)
        probe = getattr(model, "_bwd_probe", None)  # measurement hook (bench.py: fraction of all-zero adjoint rows); never set in training
        if probe is not None:
            probe(dh)
        # sigma network
        # (the backward reports max |dX| of the time-plane columns as it stores them: the field adjoint's fixed-point scale)
        pe = model.planes_encoder
        n_pl = pe.layout.n_scales * pe.layout.C
        gd_absmax = torch.zeros(1, dtype=torch.float32, device=dev) if n_pl % 16 == 0 else None  # (rows wider than 128: two launches, each reports the tiles it stores)
        dX = ops.mlp_bwd(X, act_s, dh, store.half(model.sigma_net.params), model.sigma_net.n_hidden_layers,
                         store.grad_view(model.sigma_net.params), inv, dx_absmax=gd_absmax, absmax_cols=(n_pl, 2 * n_pl))
        # field
        gcl = torch.zeros(pe.layout.numel, dtype=torch.float32, device=dev)
        fd = _field_desc(model)
        vmax = ops.absmax(pe._arena())
        hook = getattr(model, "_grads_ready_hook", None)
        # Side streams (csrc/capi.cpp): the static grid's sorted scatter and the static-plane / dynamic-hash adjoints run next to
        # the time-plane adjoint, and -- single GPU: no reducer waiting for them -- also next to the flow field's backward below,
        # which only needs d(flow).  A data-parallel trainer joins first: it starts reducing those gradients right away.
        defer = hook is None and bool(ops.streams_mask() & 2)
        res = ops.density_encode_bwd(fd, _field_grads(model, gcl), xt, flow16, tinfo, dX, inv, vmax, samples_per_ray=T, defer_join=defer,
                                     gd_absmax=gd_absmax)
        dflow16, keep = res if defer else (res, None)

        def planes_done():  # channel-last gradient arena -> added onto the planes' [1, C, H, W] gradient views, one launch
            ops.planes_relayout(pe.layout, [store.grad_view(p).view(p.shape) for p in pe._flat_planes()], gcl, to_channel_last=False,
                                accumulate=True)
        if not defer:
            planes_done()
            # every gradient except the flow field's is final now: a data-parallel trainer starts reducing them here
            if hook is not None:
                hook()
        # flow network + grid
        fn = model.flow_net
        try:
            dxf = ops.mlp_bwd(xf, act_f, dflow16, _flow_w16(model), fn.n_hidden, _flow_wgrad(model), inv)
            ops.hashgrid_t_bwd(fn.grid_enc.meta, xt, (0, 1, 2), 1, t_dev, dxf, [store.grad_view(fn.grid_enc.params)], inv)
        finally:
            if defer:  # also when a launch above raised: the side streams forked by density_encode_bwd must not stay un-joined
                ops.streams_join()  # the launch stream waits for the side streams: from here on every gradient is final in stream order
        if defer:
            del keep
            planes_done()
        pair = ctx.slice_pair
        if pair is not None:  # reference semantics for a torch optimiser: untouched slices keep .grad = None (LiDAR4D.run)
            for hd in model.hash_encoder.hash_dynamic:
                for s, enc in enumerate(hd.hash_t):
                    if s not in pair:
                        enc.params.grad = None
        return (None,) * 7 + (None,) * (len(ctx.needs_input_grad) - 7)
