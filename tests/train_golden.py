"""Loader / driver for tests/golden/train_step_losses.npz: the loss block of the reference's own ``Trainer.train_step``
(model/runner.py:166-377) run on seeded tensors by oracle/make_golden_train.py.  ``evaluate`` rebuilds the same loss from
``lidar4d_amd.trainer``'s functions; the CPU test binds the chamfer operator to the oracle, the GPU test uses the HIP kernel."""
import os

import numpy as np
import torch

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_step_losses.npz")


def cases():
    return [str(c) for c in np.load(PATH, allow_pickle=False)["cases"]]


def load(tag, device="cpu"):
    z = np.load(PATH, allow_pickle=False)
    pre = tag + "__"
    out = {}
    for k in z.files:
        if k.startswith(pre):
            v = z[k]
            out[k[len(pre):]] = torch.from_numpy(v).to(device) if v.dtype.kind in "fiub" and v.ndim > 0 else v[()]
    return out


def opt_of(c):
    o = {k[4:]: c[k] for k in c if k.startswith("opt_")}
    for k, v in o.items():
        if isinstance(v, torch.Tensor):
            o[k] = v.tolist()
        elif isinstance(v, np.generic):
            o[k] = v.item()
    return o


class FixtureFlowModel:
    """model.flow() replays the leaf tensors the reference's train_step saw, in call order (frame's points, then ground)."""

    def __init__(self, c, device):
        self.calls = 0
        self.flows = []
        for j in range(int(c["n_flow_calls"])):
            self.flows.append({k: c[f"flow{j}_{k}"].clone().to(device).requires_grad_(True) for k in ("forward", "backward")})

    def flow(self, pc, t):
        f = self.flows[self.calls]
        assert f["forward"].shape[0] == pc.shape[0]
        self.calls += 1
        return f


def evaluate(c, device="cpu", compute_loss=False):
    """-> (loss, leaves): the training loss of case ``c`` from lidar4d_amd.trainer's functions (or, compute_loss=True, from
    Trainer.compute_loss on a bare Trainer object), and the leaf tensors whose gradients the fixture holds."""
    from lidar4d_amd import trainer as T
    o = opt_of(c)
    dev = torch.device(device)
    leaf = lambda k: c[k].clone().to(dev).requires_grad_(True)
    out = {"depth_lidar": leaf("depth"), "image_lidar": leaf("image"), "weights": leaf("weights"), "z_vals": c["z_vals"].to(dev)}
    images, rays_d, time = c["images"].to(dev), c["rays_d"].to(dev), c["time"].to(dev)
    data = {"images_lidar": images, "rays_d_lidar": rays_d, "time": time}
    nf = int(o["num_frames"])
    pcs = {f"{k}": c[f"pc_{k}"].to(dev).float().contiguous() for k in range(nf)}
    grounds = {f"{k}": c[f"ground_{k}"].to(dev).float().contiguous() for k in range(nf)}
    model = FixtureFlowModel(c, dev)
    t_ground = c["flow1_t"].to(dev) if int(c["n_flow_calls"]) > 1 else None
    scale = float(o["scale"])
    gt_raydrop = images[:, :, 0]
    gt_depth = images[:, :, 2] * gt_raydrop
    if compute_loss:
        tr = object.__new__(T.Trainer)
        tr.fused_losses, tr.fused_flow_loss, tr.chamfer, tr.world = False, False, True, 1
        tr.loss_kinds = dict(depth_loss=o["depth_loss"], raydrop_loss=o["raydrop_loss"], intensity_loss=o["intensity_loss"])
        tr.flow, tr.urf, tr.iters, tr.model = bool(o["flow_loss"]), bool(o["urf_loss"]), int(o["iters"]), model
        tr.pc_list, tr.pc_ground_list = pcs, grounds
        patch = o["patch_size_lidar"]
        tr.dataset = type("D", (), dict(scale=scale, num_frames=nf, patch_size_lidar=patch))()
        tr.opt = type("O", (), dict(step_count=int(c["global_step"])))()
        if t_ground is not None:  # the reference draws torch.rand(1) for the ground points' time; replay it
            real = torch.rand
            torch.rand = lambda *a, **k: t_ground.reshape(1).to(k.get("device", "cpu"))
            try:
                loss = tr.compute_loss(data, out)
            finally:
                torch.rand = real
        else:
            loss = tr.compute_loss(data, out)
    else:
        loss = T.lidar_loss(out, images, alpha_d=o["alpha_d"], alpha_r=o["alpha_r"], alpha_i=o["alpha_i"], smooth=o["smooth_factor"],
                            depth_loss=o["depth_loss"], raydrop_loss=o["raydrop_loss"], intensity_loss=o["intensity_loss"], scale=scale)
        loss = loss + T.ray_chamfer_loss(out, data, scale)
        if o["flow_loss"]:
            loss = loss + T.flow_loss(model, pcs, grounds, time, nf, t_ground=t_ground)
        if o["urf_loss"]:
            loss = loss + T.urf_loss(out, gt_depth, int(c["global_step"]), int(o["iters"]))
        loss = loss + T.depth_grad_loss(out["depth_lidar"] * gt_raydrop, gt_depth, gt_raydrop, o["patch_size_lidar"], scale,
                                        alpha_grad=o["alpha_grad"], kind=o["depth_grad_loss"], sobel_grad=bool(o["sobel_grad"]),
                                        grad_loss=bool(o["grad_loss"]), grad_norm_smooth=bool(o["grad_norm_smooth"]),
                                        spatial_smooth=bool(o["spatial_smooth"]), tv_loss=bool(o["tv_loss"]),
                                        alpha_grad_norm=o["alpha_grad_norm"], alpha_spatial=o["alpha_spatial"], alpha_tv=o["alpha_tv"])
    leaves = {"g_depth": out["depth_lidar"], "g_image": out["image_lidar"], "g_weights": out["weights"]}
    for j, f in enumerate(model.flows):
        for k in ("forward", "backward"):
            leaves[f"flow{j}_{k}_grad"] = f[k]
    return loss, leaves


def check(c, loss, leaves, rtol=2e-5):
    want = float(c["loss"])
    got = float(loss.detach())
    assert abs(got - want) <= rtol * max(1.0, abs(want)), (got, want)
    loss.backward()
    for k, leaf in leaves.items():
        w = c[k].to(leaf.device)
        g = torch.zeros_like(leaf) if leaf.grad is None else leaf.grad
        scale = float(w.abs().max())
        assert float((g - w).abs().max()) <= rtol * max(scale, 1e-6) + 1e-9, (k, float((g - w).abs().max()), scale)
