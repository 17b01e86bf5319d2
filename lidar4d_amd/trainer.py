"""Training step around the render path: the reference's three primary losses, Adam with its two lr groups and
schedule, and ray-sharded data parallelism with one RCCL all-reduce of the flat gradient arena.

Restates only what `training rays/s` needs from the reference's Trainer (model/runner.py:166-213,474-551;
optimizer main_lidar4d.py:298-305) plus the inference step of the evaluation / simulation loops
(runner.py:438-470: staged render of a whole frame, U-Net ray-drop refinement, masking) and the optional ray-chamfer and
scene-flow and line-of-sight loss terms (runner.py:215-276), the patch depth-gradient terms (runner.py:277-367), the
optimiser-side semantics of the reference's AMP loop (GradScaler skip / backoff / growth, per-parameter Adam state:
runner.py:102,506-508) and the per-epoch parameter EMA (runner.py:534-535).  Logging is out of scope.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .params import bump_epoch


def criterion(kind, scale=1.0):
    """The element-wise loss the reference selects with --depth_loss / --intensity_loss / --raydrop_loss
    (main_lidar4d.py:63-66,183-196): l1, mse, bce (with logits) or huber (delta = 0.2 * scene scale), reduction none."""
    if kind == "l1":
        return lambda a, b: (a - b).abs()
    if kind == "mse":
        return lambda a, b: (a - b) ** 2
    if kind == "bce":
        return lambda a, b: torch.nn.functional.binary_cross_entropy_with_logits(a, b, reduction="none")
    if kind == "huber":
        return lambda a, b: torch.nn.functional.huber_loss(a, b, reduction="none", delta=0.2 * scale)
    raise ValueError(f"unknown loss criterion {kind!r} (l1, mse, bce, huber)")


def lidar_loss(outputs, images_lidar, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, smooth=0.2, depth_loss="l1",
               raydrop_loss="mse", intensity_loss="mse", scale=1.0):
    """runner.py:179-213: depth + ray-drop (label-smoothed) + intensity terms, masked by the GT ray-drop, summed.  Defaults =
    the reference's: L1 depth, MSE ray-drop, MSE intensity (main_lidar4d.py:63-66); with ``raydrop_loss='bce'`` the
    prediction goes through a sigmoid first and then BCE-with-logits, exactly as the reference does (runner.py:196-197)."""
    gt_raydrop = images_lidar[:, :, 0]
    gt_intensity = images_lidar[:, :, 1] * gt_raydrop
    gt_depth = images_lidar[:, :, 2] * gt_raydrop
    pred_raydrop = outputs["image_lidar"][:, :, 0]
    pred_intensity = outputs["image_lidar"][:, :, 1] * gt_raydrop
    pred_depth = outputs["depth_lidar"] * gt_raydrop
    if raydrop_loss == "bce":
        pred_raydrop = torch.sigmoid(pred_raydrop)
    gt_smooth = gt_raydrop.clamp(smooth, 1 - smooth)
    loss = (alpha_d * criterion(depth_loss, scale)(pred_depth, gt_depth) +
            alpha_r * criterion(raydrop_loss, scale)(pred_raydrop, gt_smooth) +
            alpha_i * criterion(intensity_loss, scale)(pred_intensity, gt_intensity))
    return loss.sum()


class _PrimaryLossFn(torch.autograd.Function):
    """``lidar_loss`` (+ ``ray_chamfer_loss / world``) with the reference's default criteria as one autograd node on
    csrc/glue.hip: forward = l4d_lidar_losses (+ l4d_chamfer_fwd, l4d_ray_chamfer_grad), which also leaves the gradients wrt
    the rendered depth / image; backward = one launch that scales them by the upstream gradient (the loss scale, a device
    scalar).  Replaces ~70 element-wise torch launches per step (runner.py:179-220 as torch evaluates it)."""

    @staticmethod
    def forward(ctx, depth, image, gt, rays_d, alpha_d, alpha_r, alpha_i, smooth, scale, chamfer, world):
        c = lambda t, *shape: t.detach().to(torch.float32).reshape(*shape).contiguous()
        n = depth.numel()
        gt_c, rd_c = c(gt, n, 3), c(rays_d, n, 3)
        loss, g_depth, g_image, pts = ops.lidar_losses(c(depth, n), c(image, n, 2), gt_c, rd_c, alpha_d, alpha_r, alpha_i, smooth, scale,
                                                       want_points=bool(chamfer))
        if chamfer:
            ops.ray_chamfer_accumulate(pts, rd_c, gt_c, 0.5 / max(n, 1) / world, scale, loss, g_depth)
        ctx.save_for_backward(g_depth, g_image)
        ctx.shapes = (depth.shape, image.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        g_depth, g_image = ctx.saved_tensors
        d, i = ops.scale_buffers(g_depth, g_image, g.detach().to(torch.float32).reshape(1).contiguous())
        return (d.view(ctx.shapes[0]), i.view(ctx.shapes[1])) + (None,) * 9


def primary_losses(outputs, data, scale, chamfer=True, world=1, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, smooth=0.2):
    """= lidar_loss(outputs, images) [+ ray_chamfer_loss(outputs, data, scale) / world] for the default criteria (L1 / MSE /
    MSE), evaluated by the fused HIP path (``_PrimaryLossFn``)."""
    return _PrimaryLossFn.apply(outputs["depth_lidar"], outputs["image_lidar"], data["images_lidar"], data["rays_d_lidar"],
                                alpha_d, alpha_r, alpha_i, smooth, scale, chamfer, world)


def frame_index(time_lidar, num_frames):
    """``int(time_lidar * (num_frames - 1))`` as the reference evaluates it (runner.py:228, lidar4d.py:143): the product is
    an fp32 TENSOR product, truncated.  (In float64 the same expression lands just below k for 26 of the 51 default frame
    times k / 50 and truncates to k - 1.)"""
    return int(np.float32(float(time_lidar)) * np.float32(num_frames - 1))


def ray_chamfer_loss(outputs, data, scale):
    """runner.py:215-220: chamfer distance between predicted and ground-truth points along the same rays (in metres),
    (dist1 + dist2).mean() * 0.5, on the HIP chamfer kernel (lidar4d_amd/chamfer.py)."""
    from .chamfer import chamfer_3DDist
    gt_raydrop = data["images_lidar"][:, :, 0]
    gt_depth = data["images_lidar"][:, :, 2] * gt_raydrop
    pred_depth = outputs["depth_lidar"] * gt_raydrop
    rays_d = data["rays_d_lidar"]
    pred_lidar = rays_d * pred_depth.unsqueeze(-1) / scale
    gt_lidar = rays_d * gt_depth.unsqueeze(-1) / scale
    dist1, dist2, _, _ = chamfer_3DDist()(pred_lidar, gt_lidar)
    return (dist1 + dist2).mean() * 0.5


def urf_loss(outputs, gt_depth, global_step, iters):
    """runner.py:255-276 (``--urf_loss``, the line-of-sight loss of Urban Radiance Fields): with a tolerance eps that
    shrinks from 0.02 to 0.002 over training, weights outside [depth - eps, depth + eps] are pushed to zero and the
    weights inside towards a (peak-normalised) Gaussian of sigma = eps / 3 around the measured depth; both terms are
    sums over all samples divided by the number of rays with a return, and enter the loss with weight 0.1.
    outputs: ``weights`` [n, T], ``z_vals`` [n, T] of the render call; gt_depth: [1, n] (already masked by ray-drop)."""
    import math
    eps = 0.02 * 0.1 ** min(global_step / iters, 1)
    weights, z = outputs["weights"], outputs["z_vals"]
    d = gt_depth.reshape(z.shape[0], 1)
    n_hit = (d > 0.0).sum()
    near = (z > d - eps) & (z < d + eps)
    empty = (z < d - eps) | (z > d + eps)
    loss_empty = ((empty * weights) ** 2).sum() / n_hit
    sigma = eps / 3.0
    bell = torch.exp(-((near * (z - d)) ** 2) / (2 * sigma ** 2)) / (sigma * math.sqrt(2 * math.pi))
    bell = bell / bell.max() * near
    loss_near = ((near * weights - bell) ** 2).sum() / n_hit
    return 0.1 * loss_empty + 0.1 * loss_near


def _patch_grads(img, sobel):
    """img [n_patch, 1, px, py] -> (d/dx, d/dy): Sobel responses (same size) or forward differences (one shorter)."""
    if sobel:
        kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32, device=img.device).view(1, 1, 3, 3)
        ky = torch.tensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]], dtype=torch.float32, device=img.device).view(1, 1, 3, 3)
        return torch.nn.functional.conv2d(img, kx, padding=1), torch.nn.functional.conv2d(img, ky, padding=1)
    return img[:, :, :, :-1] - img[:, :, :, 1:], img[:, :, :-1, :] - img[:, :, 1:, :]


def depth_grad_loss(pred_depth, gt_depth, gt_raydrop, patch_size, scale, alpha_grad=0.1, kind="l1", sobel_grad=False,
                    grad_loss=True, grad_norm_smooth=False, spatial_smooth=False, tv_loss=False, alpha_grad_norm=0.1,
                    alpha_spatial=0.1, alpha_tv=0.1):
    """runner.py:277-367: structure terms on depth PATCHES (rays drawn as px x py pixel blocks, get_lidar_rays).  The
    main term compares the horizontal depth gradient of prediction and ground truth (in metres) where the ground truth
    is smooth (|gradient| < 0.01) and has a return -- the reference only uses the x direction there, which is kept --
    summed and weighted by alpha_grad; the optional smoothness terms are means over the patch gradients.
    pred_depth / gt_depth / gt_raydrop: [1, n] in ray order (patch-major), depths already masked by the ray-drop."""
    px, py = (patch_size, patch_size) if isinstance(patch_size, int) else \
        ((patch_size[0], patch_size[0]) if len(patch_size) == 1 else tuple(patch_size))
    loss = pred_depth.new_zeros(())
    if px <= 1:
        return loss
    as_patches = lambda v: v.reshape(-1, px, py, 1).permute(0, 3, 1, 2).contiguous()
    pred = as_patches(pred_depth) / scale
    if sobel_grad:
        pgx, pgy = _patch_grads(pred, True)
    else:  # the reference takes magnitudes of the forward differences on the prediction side
        pgx, pgy = (g.abs() for g in _patch_grads(pred, False))
    dx, dy = pgx.abs(), pgy.abs()
    if grad_norm_smooth:
        loss = loss + alpha_grad_norm * (torch.exp(-dx).mean() + torch.exp(-dy).mean())
    if spatial_smooth:
        loss = loss + alpha_spatial * ((dx ** 2).mean() + (dy ** 2).mean())
    if tv_loss:
        loss = loss + alpha_tv * (dx.mean() + dy.mean())
    if grad_loss:
        gt = as_patches(gt_depth) / scale
        hit = as_patches(gt_raydrop)
        ggx, _ = _patch_grads(gt, sobel_grad)
        smooth_x = (ggx.abs() < 0.01).to(gt)
        mask = (hit if sobel_grad else hit[:, :, :, :-1]) * smooth_x
        a, b = pgx * mask, ggx * mask
        if kind == "cos":
            term = 1 - torch.nn.functional.cosine_similarity(a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1))
        elif kind == "l1":
            term = (a - b).abs()
        elif kind == "mse":
            term = (a - b) ** 2
        elif kind == "huber":
            term = torch.nn.functional.huber_loss(a, b, reduction="none", delta=0.2 * scale)
        else:
            raise ValueError(f"depth_grad_loss: unknown kind {kind!r}")
        loss = loss + alpha_grad * term.sum()
    return loss


def process_pointcloud(dataset, ground_split=None):
    """runner.py:923-951 on the device: per frame, ground-truth range image -> points (lidar4d_amd.convert) -> split into
    non-ground / ground -> scene units and world frame.  Returns (pc_list, pc_ground_list), dicts keyed by str(frame).
    The reference separates the ground with RANSAC + open3d outlier removal (utils/misc.py:128-154), which is dataset
    preprocessing and out of scope; ``ground_split(points[N,3]) -> bool mask`` stands in for it (default: within 0.15 m
    of the synthetic scene's ground plane z = -1.7 m, the reference's RANSAC distance threshold)."""
    from .convert import pano_to_lidar
    if ground_split is None:
        ground_split = lambda pts: (pts[:, 2] + 1.7).abs() < 0.15
    pc_list, pc_ground_list = {}, {}
    for k in range(dataset.num_frames):
        img = dataset.images[k]
        gt_depth = img[..., 2] * img[..., 0]
        pts = pano_to_lidar(gt_depth / dataset.scale, dataset.fov)          # metres, sensor frame
        is_ground = ground_split(pts)
        pose = dataset.poses[k]
        to_world = lambda q: (q * dataset.scale) @ pose[:3, :3].T + pose[:3, 3]
        pc_list[f"{k}"] = to_world(pts[~is_ground]).contiguous()
        pc_ground_list[f"{k}"] = to_world(pts[is_ground]).contiguous()
    return pc_list, pc_ground_list


def flow_loss(model, pc_list, pc_ground_list, time_lidar, num_frames, t_ground=None, frame_idx=None, fused=False):
    """runner.py:222-253: two-step forward / backward chamfer consistency of the scene flow between neighbouring frames'
    point clouds (sum, not mean, of the squared distances) + 0.001 * L1 of the flow on ground points at a random time.
    ``t_ground`` replaces the reference's ``torch.rand(1)`` when given (tests).  ``frame_idx``: the value of
    ``frame_index(time_lidar, num_frames)`` if the caller already knows it on the host (spares the device read-back the
    reference pays in ``int(time_lidar * ...)``)."""
    from .chamfer import chamfer_3DDist
    cham = chamfer_3DDist()
    if frame_idx is None:
        frame_idx = frame_index(time_lidar, num_frames)
    pc = pc_list[f"{frame_idx}"]
    if fused and pc.is_cuda and pc.shape[0] > 0 and hasattr(model, "_store"):  # one autograd node on csrc/glue.hip
        others = []
        for step in (1, 2):
            for sign, col0 in ((+1, 0), (-1, 3)):
                other = pc_list.get(f"{frame_idx + sign * step}")
                if other is not None and other.shape[0] > 0:
                    others.append((other.contiguous(), float(step), col0))
        ground = pc_ground_list[f"{frame_idx}"]
        if ground.shape[0] and t_ground is None:
            t_ground = torch.rand(1, device=ground.device)
        tg = None if t_ground is None else t_ground.reshape(1).to(device=pc.device, dtype=torch.float32).contiguous()
        return _SceneFlowLossFn.apply(model, pc.contiguous(), others, ground.contiguous() if ground.shape[0] else None,
                                      time_lidar.reshape(1).to(torch.float32).contiguous(), tg, *fn_params(model))
    pred = model.flow(pc, time_lidar)
    loss = pc.new_zeros(())
    for step in (1, 2):
        for sign, key in ((+1, "forward"), (-1, "backward")):
            other = pc_list.get(f"{frame_idx + sign * step}")
            if other is None or other.shape[0] == 0 or pc.shape[0] == 0:
                continue
            pc_pred = pc + pred[key].float() * step
            dist1, dist2, _, _ = cham(pc_pred.unsqueeze(0), other.unsqueeze(0))
            loss = loss + (dist1.sum() + dist2.sum()) * 0.5
    ground = pc_ground_list[f"{frame_idx}"]
    if ground.shape[0]:
        if t_ground is None:
            t_ground = torch.rand(1, device=ground.device)
        zero_flow = model.flow(ground, t_ground.reshape(1, 1).to(ground))
        loss = loss + 0.001 * (zero_flow["forward"].float().abs().sum() + zero_flow["backward"].float().abs().sum())
    return loss


class _SceneFlowLossFn(torch.autograd.Function):
    """``flow_loss`` (runner.py:222-253) as ONE autograd node for a ``LiDAR4D`` with its flat parameter store: the flow field is
    evaluated with the render path's own kernels (l4d_flow_xt -> l4d_hashgrid_t_fwd -> l4d_mlp_fwd), the warped clouds, the
    chamfer terms' sums and their gradients wrt the flow outputs come from csrc/glue.hip, and the backward runs the flow field's
    adjoint straight into the gradient arena (fp16 adjoints normalised on the device, as flow_field._FlowFn does).  About 50
    launches instead of about 165 per step.
    inputs: pc [n,3]; others [(cloud [m,3], step, col0)]; ground [ng,3] or None; t, t_ground one-element device tensors."""

    @staticmethod
    def forward(ctx, model, pc, others, ground, t_dev, t_ground, *params):
        from .fused import _flow_w16
        store, fn, dev = model._store, model.flow_net, pc.device
        w16, grid16 = _flow_w16(model), store.half(fn.grid_enc.params)
        keep_act = not ops.mlp_recompute_supported(fn.input_dim, fn.n_hidden)

        def evaluate(points, t):
            xt = ops.flow_xt(points, t, model.bound)
            xf = ops.hashgrid_t_fwd(fn.grid_enc.meta, xt, (0, 1, 2), [grid16], xt[0, 3:4], half_out=True)
            y, act = ops.mlp_fwd(xf, w16, fn.n_hidden, save_act=keep_act)
            return xt, xf, act, y

        n = pc.shape[0]
        xt, xf, act, y = evaluate(pc, t_dev)
        dy = torch.zeros(n, 6, dtype=torch.float32, device=dev)
        blocks = [(max(n, o.shape[0]) + 255) // 256 for o, _, _ in others]
        partial = torch.zeros(max(sum(blocks), 1), dtype=torch.float32, device=dev)
        if others:
            warped = ops.flow_warp(pc, y, [(col0, step) for _, step, col0 in others])
            off = 0
            for v, (other, step, col0) in enumerate(others):
                m = other.shape[0]
                dist = torch.empty(n + m, dtype=torch.float32, device=dev)
                idx = torch.empty(n + m, dtype=torch.int32, device=dev)
                ws = torch.empty(ops._lib.lib().l4d_chamfer_workspace(1, n, m), dtype=torch.uint8, device=dev)
                ops.call("l4d_chamfer_fwd", ops._p(warped[v]), ops._p(other), 1, n, m, ops._p(dist[:n]), ops._p(dist[n:]), ops._p(idx[:n]),
                         ops._p(idx[n:]), ops._p(ws), ops._stream())
                ops.call("l4d_flow_chamfer_grad", ops._p(warped[v]), n, ops._p(other), m, ops._p(dist[:n]), ops._p(dist[n:]), ops._p(idx[:n]),
                         ops._p(idx[n:]), float(step), int(col0), ops._p(dy), ops._p(partial[off:]), ops._stream())
                off += blocks[v]
        ng = 0 if ground is None else ground.shape[0]
        ev_g, dy_g, y_g = None, None, None
        if ng:
            xt_g, xf_g, act_g, y_g = evaluate(ground, t_ground)
            dy_g = torch.empty(ng, 6, dtype=torch.float32, device=dev)
            ev_g = (xt_g, xf_g, act_g)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        amax = torch.empty(2, dtype=torch.float32, device=dev)
        ops.call("l4d_flow_loss_finish", ops._p(partial), sum(blocks), ops._p(y_g), ng, 0.001, ops._p(dy_g), ops._p(dy), dy.numel(),
                 ops._p(loss), ops._p(amax), ops._stream())
        ctx.model = model
        ctx.evals = [(xt, xf, act, dy, 0)] + ([ev_g + (dy_g, 1)] if ng else [])
        ctx.amax = amax
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        from .fused import _flow_w16, _flow_wgrad
        model = ctx.model
        store, fn = model._store, model.flow_net
        store.prepare_grads()
        dev = g.device
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        w16 = _flow_w16(model)
        g_w, g_grid = _flow_wgrad(model), store.grad_view(fn.grid_enc.params)
        for xt, xf, act, dy, which in ctx.evals:
            k = dy.shape[0]
            dy16 = torch.empty(k, 16, dtype=torch.float16, device=dev)
            inv = torch.empty(1, dtype=torch.float32, device=dev)
            ops.call("l4d_flow_dy16", ops._p(dy), k, ops._p(gs), ops._p(ctx.amax[which:]), ops._p(dy16), ops._p(inv), ops._stream())
            # the adjoint kernels take their output scale from the host; the power of two is only known on the device: private
            # buffers (sparse: most entries stay zero), then y += inv * x into the arena
            gw = torch.zeros(w16.numel(), dtype=torch.float32, device=dev)
            ggrid = torch.zeros(g_grid.numel(), dtype=torch.float32, device=dev)
            dxf = ops.mlp_bwd(xf, act, dy16, w16, fn.n_hidden, gw, 1.0)
            ops.hashgrid_t_bwd(fn.grid_enc.meta, xt, (0, 1, 2), 1, xt[0, 3:4], dxf, [ggrid], 1.0)
            pending = getattr(model, "_flow_loss_pending", None)
            if pending is not None:  # (Trainer with the scene-flow term on a side stream: added to the arena once the render path's own
                pending.append((g_w, gw, g_grid, ggrid, inv))  # flow-field gradients are in -- both write the same ranges, plain stores)
            else:
                _apply_flow_grads(g_w, gw, g_grid, ggrid, inv)
        return (None,) * (6 + len(fn_params(model)))


def _apply_flow_grads(g_w, gw, g_grid, ggrid, inv):
    ops.call("l4d_axpy_dev", ops._p(g_w), ops._p(gw), gw.numel(), ops._p(inv), ops._stream())
    ops.call("l4d_axpy_dev", ops._p(g_grid), ops._p(ggrid), ggrid.numel(), ops._p(inv), ops._stream())


def fn_params(model):
    """The flow field's parameters in a fixed order (autograd inputs of _SceneFlowLossFn: grid table, then the linear layers)."""
    return [model.flow_net.grid_enc.params] + [m.weight for m in model.flow_net.linears()]


class DynamicLossScaler:
    """torch.cuda.amp.GradScaler as the reference's Trainer uses it (runner.py:102,506-508: default init_scale 65536,
    growth 2 every 2000 clean steps, backoff 0.5) with its state ON THE DEVICE -- [scale, growth tracker, found non-finite,
    1 / scale] -- so that scaling the loss, checking the reduced gradient arena (one reduction), gating the Adam launch
    and updating the scale never make the host wait (torch's GradScaler.step reads found_inf back every step).
    The scale multiplies the loss outside the fused backward, on top of ``model.loss_scale`` (= tiny-cuda-nn's constant
    internal 128); the backward kernels do not saturate fp16 adjoints, they let inf / nan travel to the
    parameter gradients (csrc/common.h f2h_grad), where ``check`` finds them."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.state = torch.tensor([init_scale, 0.0, 0.0, 1.0 / init_scale], dtype=torch.float32, device=device)

    def scale(self, loss):
        return loss * self.state[0]

    def check(self, flat_grad):
        ops.grad_nonfinite_check(flat_grad, self.state)

    def update(self):
        ops.scaler_update(self.state, self.growth_factor, self.backoff_factor, self.growth_interval)

    def get_scale(self):  # host read-back (synchronises): logging / tests only
        return float(self.state[0])

    def state_dict(self):
        st = self.state.tolist()
        return {"scale": st[0], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(st[1])}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = sd["growth_factor"], sd["backoff_factor"]
        self.growth_interval = sd["growth_interval"]
        self.state.copy_(torch.tensor([sd["scale"], float(sd["_growth_tracker"]), 0.0, 1.0 / sd["scale"]]))


class FlatAdam:
    """torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15) over the model's flat arenas in ONE l4d_adam_step_ranges launch
    (which also refreshes the fp16 compute copies); encoders at lr, networks at 0.1 lr (lidar4d.py:226-237).
    lr follows the reference's LambdaLR: lr0 * 0.1 ** min(it / iters, 1) (main_lidar4d.py:303-305).

    torch.optim.Adam keeps its state per parameter TENSOR and skips tensors whose .grad is None.  With the reference's
    ``zero_grad(set_to_none)`` that is what happens to the 6-7 of 8 HashGridT time slices a step does not touch
    (hash_field.py:79-85): no moment decay, no step increment, no movement.  Here the arena is cut into ranges -- every
    time slice table its own range, gated by the gate the backward pass sets for the slice pair it used
    (ParamStore.gates), the rest merged per lr group -- each with its own step counter on the device; a range whose gate
    is zero is left alone, and with a DynamicLossScaler the whole step is skipped when a gradient is non-finite."""

    def __init__(self, model, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, iters=30000):
        self.model, self.store = model, model._store
        self.lr0, self.betas, self.eps, self.iters = lr, betas, eps, iters
        self.group_lr = [1.0, 0.1]
        self.step_count = 0  # scheduler iterations (the reference steps its LambdaLR every iteration, skipped or not)
        self.exp_avg = torch.zeros_like(self.store.flat)
        self.exp_avg_sq = torch.zeros_like(self.store.flat)
        self._build_ranges()
        self.steps = torch.zeros(self.ranges.n, dtype=torch.int32, device=self.store.flat.device)
        self.sched = None  # device_schedule(): [iterations so far, this step's lr factor] on the device

    def device_schedule(self):
        """Evaluate the learning-rate schedule on the device from now on (l4d_adam_step_ranges ``sched``): the Adam launch then
        has no argument that changes from step to step, which is what lets a whole training step be captured into a hipGraph
        and replayed (Trainer.train_step_graphed).  ``step_count`` keeps counting on the host (checkpoints, logging)."""
        if self.sched is None:
            self.sched = torch.tensor([float(self.step_count), 1.0], dtype=torch.float32, device=self.store.flat.device)
        return self.sched

    def _build_ranges(self):
        import re
        st = self.store
        keyed = []  # (offset, group, gate) per entry in arena order
        for name, p, off, n, gi in st.entries:
            m = re.fullmatch(r"hash_encoder\.hash_dynamic\.\d+\.hash_t\.(\d+)\.params", name)
            keyed.append((off, gi, int(m.group(1)) if m else -1))
        starts = [k for i, k in enumerate(keyed) if i == 0 or k[2] >= 0 or keyed[i - 1][1:] != k[1:]]
        ends = [k[0] for k in starts[1:]] + [st.numel]
        self.ranges = ops.AdamRanges([k[0] for k in starts], [e - k[0] for k, e in zip(starts, ends)],
                                     [self.group_lr[k[1]] for k in starts], [k[2] for k in starts])
        self._range_of = {}
        for name, p, off, n, gi in st.entries:
            self._range_of[id(p)] = max(i for i, o in enumerate(self.ranges.offs) if o <= off)

    def lr(self):
        return self.lr0 * 0.1 ** min(self.step_count / self.iters, 1.0)

    def zero_grad(self):
        self.store.zero_grad() if self.store.flat_grad is not None else self.store.prepare_grads()

    # -- checkpoint interchange with torch.optim.Adam(model.get_params(lr), betas, eps) (runner.py:966,1056-1059) ----------
    def _torch_layout(self):
        """[(group index, parameter)] in the order torch.optim.Adam numbers them for LiDAR4D.get_params."""
        return [(gi, p) for gi, g in enumerate(self.model.get_params(self.lr0)) for p in g["params"]]

    def state_dict(self):
        """A state dict a ``torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)`` can load: per-parameter
        ``step`` / ``exp_avg`` / ``exp_avg_sq`` cut out of the flat buffers (parameters that never received a gradient have
        no entry, as in torch)."""
        layout = self._torch_layout()
        groups, state, lr_now = [], {}, self.lr()
        for gi, g in enumerate(self.model.get_params(self.lr0)):
            ids = [k for k, (gj, _) in enumerate(layout) if gj == gi]
            groups.append({"lr": lr_now * (g["lr"] / self.lr0), "initial_lr": g["lr"], "betas": tuple(self.betas), "eps": self.eps,
                           "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                           "differentiable": False, "fused": None, "params": ids})
        steps = self.steps.tolist()
        for k, (_, p) in enumerate(layout):
            if p.numel() == 0 or id(p) not in self.store.by_param:
                continue
            t = steps[self._range_of[id(p)]]
            if t == 0:
                continue
            off, n = self.store.by_param[id(p)]
            state[k] = {"step": torch.tensor(float(t)), "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts the optimiser state of a checkpoint written by the reference's Trainer (or by ``state_dict`` above).
        Per-parameter step counts are kept per range; tensors that share a range (planes + static grid, the networks)
        always stepped together in the reference too -- the range takes their largest count."""
        layout = self._torch_layout()
        steps = [0] * self.ranges.n
        for k, st in sd["state"].items():
            _, p = layout[int(k)]
            if p.numel() == 0 or id(p) not in self.store.by_param:
                continue
            off, n = self.store.by_param[id(p)]
            self.exp_avg[off:off + n].copy_(st["exp_avg"].to(self.exp_avg).reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].to(self.exp_avg_sq).reshape(-1))
            r = self._range_of[id(p)]
            steps[r] = max(steps[r], int(float(st["step"])))
        self.steps.copy_(torch.tensor(steps, dtype=torch.int32))
        self.step_count = max(steps) if steps else 0  # checkpoint.load_checkpoint overrides it with the scheduler's count
        self.sync_device_schedule()

    def sync_device_schedule(self):
        """The device-side LR schedule ([iterations, factor], created by the first graphed step) follows ``step_count`` again --
        after a checkpoint load the old iteration count would otherwise keep driving the learning rate."""
        if self.sched is not None:
            self.sched.copy_(torch.tensor([float(self.step_count), 0.1 ** min(self.step_count / self.iters, 1.0)]))

    def step(self, grad_scale=1.0, scaler=None):
        st = self.store
        lr = self.lr()
        self.step_count += 1
        f16 = st.refresh16()  # no-op while current; after load_state_dict / EMA copy_to the gated-off ranges would otherwise stay stale
        if self.steps.device != st.flat.device:
            self.steps = self.steps.to(st.flat.device)
        if self.sched is not None:
            lr = self.lr0  # the factor 0.1 ** min(it / iters, 1) is applied on the device
        ops.adam_step_ranges(st.flat, st.flat_grad, self.exp_avg, self.exp_avg_sq, f16, self.ranges, lr, st.gates,
                             None if scaler is None else scaler.state, self.steps, self.betas[0], self.betas[1], self.eps,
                             grad_scale, sched=self.sched, sched_iters=float(self.iters))
        bump_epoch()          # parameters changed behind torch's version counters ...
        st.mark16_current()   # ... and the fp16 copies were refreshed by the same kernel
        self.model.planes_encoder._cl_key = None  # channel-last plane copy must be rebuilt


def refine_unet(unet, raydrop_input, raydrop_gt, epochs=1000, batch_size=None, lr=0.001, box_num_max=32, rng=None, log=None):
    """The ray-drop refinement stage, runner.py:865-912: train the U-Net on the stacked [ray-drop, intensity, depth] images
    the frozen field renders for the training frames (raydrop_input [B, 3, H, W]) against the ground-truth ray-drop masks
    (raydrop_gt [B, 1, H, W]) -- BCE loss, Adam(lr) under a OneCycleLR of ``epochs`` steps, and per step a random number
    (< box_num_max) of rectangles, each up to 10 % of the image per side, blanked out of all input channels so that the
    network learns to fill holes.  ``rng``: numpy Generator / RandomState for the boxes and the batch choice (the reference
    draws from the global numpy state).  Returns the list of losses.  Plain torch modules: runs on MIOpen on the GPU."""
    rng = rng if rng is not None else np.random
    randint = (lambda lo, hi=None: int(rng.integers(lo, hi) if hi is not None else rng.integers(lo))) if hasattr(rng, "integers") \
        else (lambda lo, hi=None: int(rng.randint(lo, hi) if hi is not None else rng.randint(lo)))
    unet.train()
    optimizer = torch.optim.Adam(unet.parameters(), lr=lr, weight_decay=0)
    scheduler = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=lr, total_steps=epochs)
    bce = torch.nn.BCELoss()
    H, W = raydrop_input.shape[2], raydrop_input.shape[3]
    max_h, max_w = int(0.1 * H), int(0.1 * W)
    losses = []
    for it in range(epochs):
        optimizer.zero_grad()
        if batch_size is not None:
            pick = torch.as_tensor(rng.choice(raydrop_input.shape[0], batch_size, replace=False), device=raydrop_input.device)
            x, gt = raydrop_input[pick], raydrop_gt[pick]
        else:
            x, gt = raydrop_input, raydrop_gt
        keep = torch.ones_like(x)
        for _ in range(randint(box_num_max)):
            bh, bw = randint(1, max(max_h, 2)), randint(1, max(max_w, 2))
            y0, x0 = randint(H - bh), randint(W - bw)
            keep[:, :, y0:y0 + bh, x0:x0 + bw] = 0.0
        loss = bce(unet(x * keep), gt)
        loss.backward()
        losses.append(float(loss.detach()))
        if log is not None and it % 50 == 0:
            log(f"iter:{it}, lr:{optimizer.param_groups[0]['lr']:.6f}, raydrop loss:{losses[-1]}")
        optimizer.step()
        scheduler.step()
    unet.eval()
    return losses


class FlatEMA:
    """Exponential moving average of the parameters over the flat arena: what the reference keeps with
    ``torch_ema.ExponentialMovingAverage(model.parameters(), decay)`` (runner.py:97-98; updated ONCE PER EPOCH, after the
    loop over the loader, :534-535; swapped in for evaluation with store / copy_to / restore, :565-567,679-680; ``--ema_decay`` 0.95 by
    default).  One fused elementwise launch over one buffer instead of a python loop over ~70 tensors.
    Update rule (torch_ema): decay_t = min(decay, (1 + t) / (10 + t)) for the t-th update; shadow -= (1 - decay_t) *
    (shadow - param).  The U-Net is not part of the arena and is left alone (the reference trains it after the field)."""

    def __init__(self, model, decay=0.95, use_num_updates=True):
        self.model, self._st = model, model._store
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow = self._st.flat.detach().clone()
        self.backup = None

    def update(self):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        self.shadow.lerp_(self._st.flat.detach(), 1.0 - decay)

    def _write(self, src):
        with torch.no_grad():
            self._st.flat.copy_(src)
        bump_epoch()                                  # fp16 compute copies must be refreshed
        self.model.planes_encoder._cl_key = None      # and the channel-last plane copy

    def store(self):
        self.backup = self._st.flat.detach().clone()

    def copy_to(self):
        self._write(self.shadow)

    def restore(self):
        if self.backup is None:
            raise RuntimeError("FlatEMA.restore() without a preceding store()")
        self._write(self.backup)
        self.backup = None

    def _views(self, flat):
        return [flat[off:off + n].view(p.shape) for _, p, off, n, _ in self._st.entries]

    def state_dict(self):
        """torch_ema's keys.  ``shadow_params`` follows ``model.parameters()`` like torch_ema's list does (the registration
        order equals the reference's: tests/golden/param_order.npz); parameters outside the arena (the zero-size
        Frequency-encoding tensor, the U-Net) appear with their current values."""
        shadows = []
        for p in self.model.parameters():
            if id(p) in self._st.by_param and p.numel():
                off, n = self._st.by_param[id(p)]
                shadows.append(self.shadow[off:off + n].view(p.shape).clone())
            else:
                shadows.append(p.detach().clone())
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": shadows, "collected_params": None}

    def load_state_dict(self, sd):
        params = list(self.model.parameters())
        if len(sd["shadow_params"]) != len(params):
            raise ValueError(f"FlatEMA: {len(sd['shadow_params'])} shadow tensors for {len(params)} model parameters")
        self.decay, self.num_updates = sd["decay"], sd["num_updates"]
        for p, t in zip(params, sd["shadow_params"]):
            if id(p) in self._st.by_param and p.numel():
                off, n = self._st.by_param[id(p)]
                self.shadow[off:off + n].copy_(t.to(self.shadow).reshape(-1))


class GradReducer:
    """Data-parallel SUM all-reduce of the flat gradient arena in two phases, so that most of it hides behind the tail
    of the backward pass: the fused backward produces the gradients in the order attribute nets -> sigma net -> planes /
    hash tables -> flow MLP -> flow grid, and calls ``early()`` (through ``model._grads_ready_hook``) when everything
    but the flow field's range is final.  Those two ranges (about 2/3 of the 186 MB) are reduced asynchronously on
    RCCL's stream while the flow-field adjoint (several ms of kernels) still runs; ``finish()`` reduces the flow range
    and waits.  Without a preceding ``early()`` it falls back to one all-reduce of the whole arena."""

    def __init__(self, model, transport=None):
        """transport: "fp32" (default: plain SUM all-reduce) or "bf16" (``Trainer(grad_transport="bf16")``): the encoder range --
        planes + hash tables, 97 % of the bytes -- travels as bf16 with fp32 accumulation on arrival: every rank sends chunk r
        of its gradient to rank r (all-to-all), sums the world_size chunks it received in fp32, and the reduced chunks are
        all-gathered as bf16.  Same wire pattern as a ring all-reduce (reduce-scatter + all-gather) at half the bytes per
        xGMI link; the reduced value is rounded to bf16 once (2^-9 relative: noise Adam's normalisation tolerates; inf / nan
        survive, so the overflow check still works).  The network ranges and the gates stay fp32."""
        self.store = st = model._store
        self.transport = transport or "fp32"
        if self.transport not in ("fp32", "bf16"):
            raise ValueError(f"GradReducer: unknown transport {self.transport!r}")
        flow = [(off, n) for name, _, off, n, _ in st.entries if name.startswith("flow_net.")]
        after = [off for name, _, off, n, _ in st.entries if name.startswith("sigma_net.")]
        self.flow_lo = min(off for off, _ in flow)
        self.flow_hi = min(after)  # flow_net is the first block of lr group 1 (lidar4d.py:226-237 order); sigma_net follows
        assert self.flow_lo == st.group_ranges[1][0] and all(self.flow_lo <= off < self.flow_hi for off, _ in flow)
        self.works = []
        self._pending16 = None
        # what a scaling record needs to show how much of the collective was hidden (bench.py): bytes on the wire per phase, and -- when
        # ``record_timing`` is set -- the time the launch stream spent inside finish() (HIP events, read back after the timed region)
        wire = 2 if self.transport == "bf16" else 4
        self.bytes_early = self.flow_lo * wire + (st.grad_numel - self.flow_hi) * 4  # encoders (+ networks and gates: always fp32)
        self.bytes_late = (self.flow_hi - self.flow_lo) * 4                          # the flow field's range
        self.record_timing = False
        self._timing = []

    def exposed_wait_ms(self):
        """Mean time per step the launch stream was held in finish() -- the flow range's all-reduce plus whatever of the early phase
        had not completed under the flow field's adjoint -- over the steps recorded since ``record_timing`` was set."""
        if not self._timing:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._timing]
        self._timing = []
        return sum(ms) / len(ms)

    # -- bf16 transport of one range -----------------------------------------------------------------------------------
    def _start16(self, lo, hi):
        g = self.store.flat_grad
        world = dist.get_world_size()
        n = hi - lo
        chunk = (n + world - 1) // world
        send = torch.zeros(world * chunk, dtype=torch.bfloat16, device=g.device)
        send[:n].copy_(g[lo:hi])
        recv = torch.empty_like(send)
        work = dist.all_to_all_single(recv, send, async_op=True)
        self._pending16 = (lo, hi, chunk, world, send, recv, work)

    def _finish16(self):
        lo, hi, chunk, world, send, recv, work = self._pending16
        work.wait()
        mine = recv.view(world, chunk).float().sum(0)  # fp32 accumulation on arrival
        out = torch.empty(world * chunk, dtype=torch.bfloat16, device=recv.device)
        dist.all_gather_into_tensor(out, mine.to(torch.bfloat16))
        self.store.flat_grad[lo:hi].copy_(out[:hi - lo])
        self._pending16 = None

    def early(self):
        g = self.store.flat_grad
        if self.transport == "bf16":
            self._start16(0, self.flow_lo)
            self.works = [dist.all_reduce(g[self.flow_hi:self.store.grad_numel], op=dist.ReduceOp.SUM, async_op=True)]
            return
        self.works = [dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True)
                      for a, b in ((0, self.flow_lo), (self.flow_hi, self.store.grad_numel)) if b > a]  # incl. the gates

    def finish(self):
        if self.record_timing and self.store.flat_grad.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._finish()
            b.record()
            self._timing.append((a, b))
        else:
            self._finish()

    def _finish(self):
        g = self.store.flat_grad
        if not self.works:
            if self.transport == "bf16":
                self._start16(0, self.flow_lo)
                dist.all_reduce(g[self.flow_lo:self.store.grad_numel], op=dist.ReduceOp.SUM)
                self._finish16()
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return
        dist.all_reduce(g[self.flow_lo:self.flow_hi], op=dist.ReduceOp.SUM)
        for w in self.works:
            w.wait()
        self.works = []
        if self._pending16 is not None:
            self._finish16()


class Trainer:
    """One process per GPU.  Rays (whole frames) are sharded across ranks -- each rank draws its own frame and ray
    indices -- parameters are replicated and the flat gradient buffer is SUM-all-reduced once per step (the primary
    loss is a sum over rays, so the result equals one big batch)."""

    def __init__(self, model, dataset, lr=1e-2, iters=30000, num_steps=768, chamfer=True, flow=True, urf=False,
                 ema_decay=None, loss_scaler=True, init_scale=65536.0, depth_loss="l1", raydrop_loss="mse",
                 intensity_loss="mse", epoch_steps=None, fused_losses=True, force_allreduce=False, overlap_allreduce=True,
                 grad_transport="fp32", graph_batch_inside=True, flow_loss_stream=True):
        """Defaults follow the reference's default run: the ray chamfer term is always part of its step
        (runner.py:215-220) and ``--flow_loss`` defaults to True (main_lidar4d.py:67).
        chamfer: a mean over the rank's own rays, so under data parallelism it is scaled by 1/world before the SUM
        all-reduce.  flow: the scene-flow consistency term (runner.py:222-253), a per-frame sum, enters each
        rank's loss as it is (every rank works on its own frame).  loss_scaler: GradScaler semantics on the device
        (DynamicLossScaler).  ema_decay: parameter EMA, updated once per epoch like the reference's (runner.py:534-535);
        an epoch = ``epoch_steps`` steps (default: one per training frame, the reference's loader length)."""
        self.model, self.dataset, self.num_steps, self.chamfer = model, dataset, num_steps, chamfer
        self.flow, self.urf, self.iters = flow, urf, iters
        self.loss_kinds = dict(depth_loss=depth_loss, raydrop_loss=raydrop_loss, intensity_loss=intensity_loss)
        # the default criteria run as one fused node (primary_losses); any other choice, or fused_losses=False, takes the torch
        # restatement of runner.py:179-220 (lidar_loss / ray_chamfer_loss; what the tests compare the fused nodes with)
        self.fused_losses = (depth_loss, raydrop_loss, intensity_loss) == ("l1", "mse", "mse") and bool(fused_losses)
        self.fused_flow_loss = bool(fused_losses)  # the scene-flow term as one autograd node (_SceneFlowLossFn)
        self.graph_batch_inside = bool(graph_batch_inside)
        # flow_loss_stream: the scene-flow term (about 50 small launches on a frame's point clouds, none of which fills the chip) runs
        # on a stream of its own next to the render path's forward and backward (eager steps only; a captured step keeps one stream)
        self.flow_loss_stream = bool(flow_loss_stream)
        self._flow_stream = None
        self.ema = FlatEMA(model, ema_decay) if ema_decay is not None else None  # runner.py:97-98
        self.epoch_steps = epoch_steps if epoch_steps is not None else getattr(dataset, "num_frames", 1)
        self.local_step = 0
        if flow:
            self.pc_list, self.pc_ground_list = process_pointcloud(dataset)
        self.opt = FlatAdam(model, lr=lr, iters=iters)
        self.scaler = DynamicLossScaler(model._store.flat.device, init_scale=init_scale) if loss_scaler else None
        model.reference_grad_none = False  # untouched time slices are gated on the device (FlatAdam), no host read-back
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # a one-rank process group still runs the collective when asked to (bench.py --force-dist: exercises RCCL on one GPU)
        self.force_allreduce = dist.is_available() and dist.is_initialized() and bool(force_allreduce)
        self.reducer = None
        if self.world > 1 or self.force_allreduce:
            self.reducer = GradReducer(model, transport=grad_transport)
            if overlap_allreduce:  # (False: one all-reduce of the whole arena behind the backward pass)
                model._grads_ready_hook = self.reducer.early

    def compute_loss(self, data, out, flow_term=None, t_ground=None):
        """The reference's training loss (runner.py:179-276,277-367) for one batch and its render outputs.
        flow_term: the scene-flow term if the caller has evaluated it already (train_step on a side stream)."""
        if self.fused_losses and out["depth_lidar"].is_cuda:
            loss = primary_losses(out, data, self.dataset.scale, chamfer=self.chamfer, world=self.world)
        else:
            loss = lidar_loss(out, data["images_lidar"], scale=self.dataset.scale, **self.loss_kinds)
            if self.chamfer:
                loss = loss + ray_chamfer_loss(out, data, self.dataset.scale) / self.world
        if self.flow and flow_term is not None:
            loss = loss + flow_term
        elif self.flow:
            loss = loss + self._flow_term(data, t_ground)
        patch = getattr(self.dataset, "patch_size_lidar", 1)
        if patch != 1:  # rays were drawn as pixel patches (runner.py:277-367); a sum over this rank's patches
            gt = data["images_lidar"]
            loss = loss + depth_grad_loss(out["depth_lidar"] * gt[:, :, 0], gt[:, :, 2] * gt[:, :, 0], gt[:, :, 0], patch,
                                          self.dataset.scale)
        if self.urf:  # a per-ray mean like the chamfer term
            gt = data["images_lidar"]
            loss = loss + urf_loss(out, gt[:, :, 2] * gt[:, :, 0], self.opt.step_count, self.iters) / self.world
        return loss

    def _flow_term(self, data, t_ground=None):
        known = frame_index(data["time_host"], self.dataset.num_frames) if "time_host" in data else None
        return flow_loss(self.model, self.pc_list, self.pc_ground_list, data["time"], self.dataset.num_frames, frame_idx=known,
                         fused=self.fused_flow_loss, t_ground=t_ground)

    def train_step(self, data=None):
        data = data if data is not None else self.dataset.batch()
        loss = self._step_device_work(data)
        self._step_host_bookkeeping()
        return loss

    def _step_host_bookkeeping(self):
        self.local_step += 1
        if self.local_step % self.epoch_steps == 0:
            self.end_epoch()

    # -- the same step as ONE hipGraph per frame -------------------------------------------------------------------------
    def graphs_supported(self):
        """A step can be captured when nothing in it needs the host: single rank (the RCCL all-reduce is issued by torch's
        process group), a dataset that draws its batch on the device (``batch_for`` + ``register`` of its generator), no
        patch / line-of-sight terms with host-side schedules."""
        return (self.reducer is None and not self.urf and hasattr(self.dataset, "batch_for") and hasattr(self.dataset, "next_frame")
                and getattr(self.dataset, "patch_size_lidar", 1) == 1 and self.model._store.flat.is_cuda)

    def train_step_graphed(self, frame=None):
        """train_step as the replay of a hipGraph captured per frame index (the scene-flow loss walks that frame's point clouds, so
        the launch sequence depends on the frame and on nothing else): batch draw (device RNG), forward, losses, backward,
        GradScaler check / skip / update, Adam with the learning-rate schedule on the device -- about 640 launches per step
        become one.  The first call for a frame runs one eager step (refreshes every host-side cache) and captures the next."""
        if not self.graphs_supported():
            raise RuntimeError("Trainer.train_step_graphed: this configuration needs the host inside a step (see graphs_supported)")
        if frame is None:
            frame = self.dataset.next_frame()
        st = self.__dict__.setdefault("_step_graphs", {"pool": None, "graphs": {}})
        self.opt.device_schedule()
        # graph_batch_inside=False: the batch is drawn by eager launches into static buffers before every replay (debugging aid,
        # tests/test_gpu_optim.py; default: the draw is part of the graph, the dataset's device generator is registered with it)
        outside = not self.graph_batch_inside
        rec = st["graphs"].get(frame)
        if rec is None:
            static = None
            if outside:
                static = {k: (v.contiguous().clone() if torch.is_tensor(v) else v) for k, v in self.dataset.batch_for(frame).items()}
            draw = (lambda: static) if outside else (lambda: self.dataset.batch_for(frame))
            # torch's capture recipe: the warm-up step runs on the (non-default) stream the capture will use, and nothing of
            # its autograd graph survives it -- an AccumulateGrad node that was created on the legacy default stream and is
            # still alive makes the captured backward synchronise with that stream, which a capture cannot contain.  The
            # library's side streams, when switched on, fork and join by events and are captured like any other dependency.
            side = st.setdefault("stream", torch.cuda.Stream())
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                loss = self._step_device_work(draw()).detach()  # eager: leaves every cache in its steady state
            torch.cuda.current_stream().wait_stream(side)
            self.opt.step_count -= 1  # (the capture below is not executed: it must not count as an iteration on the host)
            graph = torch.cuda.CUDAGraph()
            gen = getattr(self.dataset, "gen", None)
            if not outside and gen is not None and hasattr(graph, "register_generator_state"):
                graph.register_generator_state(gen)
            with torch.cuda.graph(graph, pool=st["pool"], stream=side):  # (one memory pool for all frames' graphs: they never overlap)
                loss_g = self._step_device_work(draw()).detach()
            if st["pool"] is None:
                st["pool"] = graph.pool()
            st["graphs"][frame] = {"graph": graph, "loss": loss_g, "static": static}
            self._step_host_bookkeeping()
            return loss
        if rec["static"] is not None:
            for k, v in self.dataset.batch_for(frame).items():
                if torch.is_tensor(v):
                    rec["static"][k].copy_(v)
        # The captured step holds no fp32 -> fp16 refresh (it was captured right after an Adam step, when the copy was current);
        # parameters written outside the graph (EMA copy_to / restore, load_state_dict) are cast here -- a host no-op while current.
        self.model._store.refresh16()
        rec["graph"].replay()
        self.opt.step_count += 1
        self._step_host_bookkeeping()
        return rec["loss"]

    def _step_device_work(self, data):
        self.opt.zero_grad()
        st = self.model._store
        side, flow_term = None, None
        # the ground points' random time of the scene-flow term (runner.py:247) is drawn here, in front of the render's sample jitter,
        # whichever stream the term then runs on: one random stream for both forms of the step
        t_ground = torch.rand(1, device=st.flat.device) if self.flow else None
        try:
            if (self.flow and self.flow_loss_stream and self.fused_flow_loss and st.flat.is_cuda
                    and not torch.cuda.is_current_stream_capturing()):
                # the scene-flow term on its own stream: forward next to the render forward, backward (autograd runs a node on the
                # stream of its forward) next to the render backward; its parameter gradients wait in private buffers
                # (model._flow_loss_pending).  Measured at C3: 30.19 -> 29.76 ms per step (session s19 of round 6).
                main = torch.cuda.current_stream()
                side = self._flow_stream = self._flow_stream or torch.cuda.Stream()
                side.wait_stream(main)  # (the gradient fill above, last step's Adam: parameters and fp16 copies are current)
                self.model._flow_loss_pending = []
                with torch.cuda.stream(side):
                    flow_term = self._flow_term(data, t_ground)
            out = self.model.render(data["rays_o_lidar"], data["rays_d_lidar"], data["time"], staged=False, perturb=True,
                                    num_steps=self.num_steps, time_host=data.get("time_host"))
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                flow_term.record_stream(torch.cuda.current_stream())
            loss = self.compute_loss(data, out, flow_term=flow_term, t_ground=t_ground)
            (self.scaler.scale(loss) if self.scaler is not None else loss).backward()  # runner.py:506
            if side is not None:  # the render path's flow-field gradients are in the arena: add the scene-flow term's
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for args in self.model._flow_loss_pending:
                        _apply_flow_grads(*args)
                torch.cuda.current_stream().wait_stream(side)
        finally:
            if side is not None:
                self.model._flow_loss_pending = None
        if self.flow:
            st.prepare_grads()  # fold gradients autograd produced outside the fused node into the arena
        if self.reducer is not None:
            self.reducer.finish()
        if self.scaler is not None:
            self.scaler.check(st.flat_grad)      # after the all-reduce: every rank takes the same decision
        self.opt.step(scaler=self.scaler)        # runner.py:507 (skipped on the device if a gradient was non-finite)
        if self.scaler is not None:
            self.scaler.update()                 # runner.py:508
        return loss

    def end_epoch(self):
        """runner.py:534-535: the parameter EMA is updated once per epoch, after the loop over the loader."""
        if self.ema is not None:
            self.ema.update()

    @torch.no_grad()
    def collect_refine_data(self, frames=None, max_ray_batch=4096):
        """runner.py:824-863: render every training frame with the (frozen) field and stack what the U-Net sees --
        returns (raydrop_input [B, 3, H, W], raydrop_gt [B, 1, H, W]) for ``refine_unet``."""
        self.model.eval()
        if self.ema is not None:
            self.ema.copy_to()  # the reference refines on the EMA weights and drops the raw ones (runner.py:819-821)
        inputs, gts = [], []
        for k in (range(self.dataset.num_frames) if frames is None else frames):
            fr = self.dataset.frame(k)
            H, W = fr["H_lidar"], fr["W_lidar"]
            out = self.model.render(fr["rays_o_lidar"], fr["rays_d_lidar"], fr["time"], staged=True, perturb=False,
                                    max_ray_batch=max_ray_batch, num_steps=self.num_steps)
            image = out["image_lidar"].reshape(-1, H, W, 2)
            inputs.append(torch.cat([image[..., 0], image[..., 1], out["depth_lidar"].reshape(-1, H, W)], dim=0).unsqueeze(0))
            gts.append(fr["images_lidar"][:, :, :, 0].unsqueeze(0))
        return torch.cat(inputs, 0).float().contiguous(), torch.cat(gts, 0).float().contiguous()

    @torch.no_grad()
    def test_step(self, data, refine=True, perturb=False, max_ray_batch=4096, raydrop_loss="mse", alpha_r=0.01):
        """runner.py:438-470: render every ray of a frame in chunks (staged), refine the ray-drop probability with the
        U-Net on the stacked [ray-drop, intensity, depth] image, mask intensity and depth by ray-drop > 0.5.
        data: ``rays_o_lidar``/``rays_d_lidar`` [B, H*W, 3], ``time`` [B,1], ``H_lidar``, ``W_lidar``.
        Returns (pred_raydrop [1,H,W], pred_intensity [B,H,W], pred_depth [B,H,W])."""
        H, W = data["H_lidar"], data["W_lidar"]
        out = self.model.render(data["rays_o_lidar"], data["rays_d_lidar"], data["time"], staged=True, perturb=perturb,
                                max_ray_batch=max_ray_batch, num_steps=self.num_steps)
        image = out["image_lidar"].reshape(-1, H, W, 2)
        pred_raydrop, pred_intensity = image[..., 0], image[..., 1]
        pred_depth = out["depth_lidar"].reshape(-1, H, W)
        if raydrop_loss == "bce":
            pred_raydrop = torch.sigmoid(pred_raydrop)
        if refine:
            stacked = torch.cat([pred_raydrop, pred_intensity, pred_depth], dim=0).unsqueeze(0)
            pred_raydrop = self.model.unet(stacked).squeeze(0)
        raydrop_mask = torch.where(pred_raydrop > 0.5, 1, 0)
        if alpha_r > 0:
            pred_intensity = pred_intensity * raydrop_mask
            pred_depth = pred_depth * raydrop_mask
        return pred_raydrop, pred_intensity, pred_depth
