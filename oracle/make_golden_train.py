"""Generate tests/golden/train_step_losses.npz by running the loss block of the REAL reference's ``Trainer.train_step``
(model/runner.py:166-377).  Build container only.

    python -m oracle.make_golden_train      # needs /root/reference; writes tests/golden/train_step_losses.npz

``model/runner.py`` imports cv2, imageio, tensorboardX, torch_ema, rich and the chamfer CUDA extension at module level; none of
them is installed here and none of them takes part in the loss arithmetic, so empty stand-in MODULE OBJECTS are bound to those
names while the reference's file is imported from a throw-away scratch copy (the ``tinycudann`` pattern of make_golden.py;
nothing of the reference is copied into this repo).  The one stand-in that computes is ``chamfer_3DDist``: the CUDA kernel
cannot run without a GPU, so the brute-force restatement ``oracle.chamfer_ref`` is bound in its place -- the chamfer VALUES stay
"parity unpinned" (oracle/chamfer_ref.py), everything train_step does with them is the reference's own code.

``train_step`` is run as an unbound method on a bare object that carries exactly the attributes the loss block reads
(``opt``, ``model``, ``criterion`` -- the table of main_lidar4d.py:183-196 --, ``cham_fn``, ``pc_list``, ``pc_ground_list``,
``global_step``, ``device``).  ``model.render`` / ``model.flow`` return seeded leaf tensors, so the fixture holds, per case: the
inputs, the options, the loss value and d(loss) / d(every model output).  tests/test_host_logic.py checks
``lidar4d_amd.trainer``'s loss functions against it on the CPU, tests/test_gpu_glue.py the fused HIP losses on the GPU.
"""
import argparse
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

from oracle import chamfer_ref
from oracle.make_golden import REF, save

STUBS = ("cv2", "imageio", "tensorboardX", "torch_ema", "rich", "rich.console", "lpips", "open3d",
         "utils.chamfer3D", "utils.chamfer3D.dist_chamfer_3D", "utils.misc")


class _Chamfer(torch.nn.Module):
    def forward(self, a, b):
        return chamfer_ref.chamfer(a, b)


def import_reference_trainer():
    scratch = tempfile.mkdtemp(prefix="l4d_ref_train_")
    dst = os.path.join(scratch, "ref")
    shutil.copytree(REF, dst)
    for name in STUBS:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["rich.console"].Console = object
    sys.modules["torch_ema"].ExponentialMovingAverage = object
    sys.modules["utils.chamfer3D.dist_chamfer_3D"].chamfer_3DDist = _Chamfer
    sys.modules["utils.misc"].point_removal = None
    sys.path.insert(0, dst)
    import model.runner as ref_runner  # noqa
    return ref_runner.Trainer, scratch


def criterion_table(opt):
    """the dict main_lidar4d.py:183-196 builds (call-site configuration, not part of runner.py)"""
    loss_dict = {"mse": torch.nn.MSELoss(reduction="none"), "l1": torch.nn.L1Loss(reduction="none"),
                 "bce": torch.nn.BCEWithLogitsLoss(reduction="none"),
                 "huber": torch.nn.HuberLoss(reduction="none", delta=0.2 * opt.scale), "cos": torch.nn.CosineSimilarity()}
    return {"depth": loss_dict[opt.depth_loss], "raydrop": loss_dict[opt.raydrop_loss],
            "intensity": loss_dict[opt.intensity_loss], "grad": loss_dict[opt.depth_grad_loss]}


class StubModel:
    """render() / flow() hand out seeded LEAF tensors and remember them: the fixture stores d(loss) / d(each)."""

    def __init__(self, n, T, gen, scale, gt_depth):
        u = lambda *s: torch.rand(*s, generator=gen)
        # depth near the ground truth (so that the line-of-sight masks select both sides), image in (0, 1)
        self.depth = (gt_depth + (u(1, n) - 0.5) * 4 * scale).clamp_min(0.0).requires_grad_(True)
        self.image = u(1, n, 2).requires_grad_(True)
        z = torch.linspace(0.0, 1.0, T).unsqueeze(0) * 81 * scale + (u(n, T) - 0.5) * scale
        self.z_vals = z
        self.weights = (u(n, T) * (u(n, T) < 0.1)).requires_grad_(True)
        self.flows = []
        self.flow_t = []
        self.gen = gen

    def render(self, rays_o, rays_d, time, **kw):
        self.render_kw = {k: kw[k] for k in ("staged", "perturb", "force_all_rays")}
        return {"depth_lidar": self.depth, "image_lidar": self.image, "weights": self.weights, "z_vals": self.z_vals}

    def flow(self, pc, t):
        f = {k: ((torch.rand(pc.shape[0], 3, generator=self.gen) - 0.5) * 0.02).requires_grad_(True) for k in ("forward", "backward")}
        self.flows.append(f)
        self.flow_t.append(t.detach().clone().reshape(-1))
        return f


def run_case(Trainer, tag, n, T, seed, frame, **optkw):
    gen = torch.Generator().manual_seed(seed)
    scale = 0.010504329815187737
    opt = argparse.Namespace(patch_size_lidar=1, raydrop_loss="mse", depth_loss="l1", intensity_loss="mse", depth_grad_loss="l1",
                             smooth_factor=0.2, alpha_d=1.0, alpha_r=0.01, alpha_i=0.1, scale=scale, flow_loss=False, num_frames=5,
                             urf_loss=False, iters=1000, sobel_grad=False, grad_norm_smooth=False, spatial_smooth=False, tv_loss=False,
                             grad_loss=False, alpha_grad=0.1, alpha_grad_norm=0.1, alpha_spatial=0.1, alpha_tv=0.1)
    for k, v in optkw.items():
        setattr(opt, k, v)
    u = lambda *s: torch.rand(*s, generator=gen)
    images = torch.stack([(u(1, n) > 0.25).float(), u(1, n), (4.0 + 60.0 * u(1, n)) * scale], -1)  # raydrop, intensity, depth
    d = torch.nn.functional.normalize(u(1, n, 3) - 0.5, dim=-1)
    o = (u(1, 1, 3) - 0.5).expand(1, n, 3) * 0.01
    time = torch.tensor([[frame / (opt.num_frames - 1)]], dtype=torch.float32)
    pcs = {f"{k}": ((u(40 + 7 * k, 3) - 0.5) * 0.6).numpy() for k in range(opt.num_frames)}
    grounds = {f"{k}": ((u(11 + k, 3) - 0.5) * 0.6).numpy() for k in range(opt.num_frames)}
    gt_depth = images[:, :, 2] * images[:, :, 0]
    model = StubModel(n, T, gen, scale, gt_depth)

    tr = object.__new__(Trainer)
    tr.opt, tr.model, tr.criterion, tr.cham_fn = opt, model, criterion_table(opt), _Chamfer()
    tr.pc_list, tr.pc_ground_list, tr.global_step, tr.device, tr.log_ptr = pcs, grounds, 250, torch.device("cpu"), None
    data = {"rays_o_lidar": o, "rays_d_lidar": d, "time": time, "images_lidar": images}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # train_step moves the frame's point clouds to "the GPU"
    try:
        pred_i, gt_i, pred_d, gt_d, loss = Trainer.train_step(tr, data)
    finally:
        torch.Tensor.cuda = real_cuda
    loss.backward()
    z = lambda t, like: torch.zeros_like(like) if t is None else t
    out = dict(n=n, T=T, frame=frame, global_step=tr.global_step, images=images, rays_o=o, rays_d=d, time=time,
               depth=model.depth.detach(), image=model.image.detach(), weights=model.weights.detach(), z_vals=model.z_vals,
               loss=loss.detach(), g_depth=z(model.depth.grad, model.depth), g_image=z(model.image.grad, model.image),
               g_weights=z(model.weights.grad, model.weights), pred_depth_ret=pred_d.detach(), gt_depth_ret=gt_d.detach(),
               render_perturb=model.render_kw["perturb"], render_staged=model.render_kw["staged"],
               render_force_all_rays=model.render_kw["force_all_rays"], n_flow_calls=len(model.flows))
    for k, v in vars(opt).items():
        out["opt_" + k] = np.asarray(v)
    for k in range(opt.num_frames):
        out[f"pc_{k}"], out[f"ground_{k}"] = pcs[f"{k}"], grounds[f"{k}"]
    for j, f in enumerate(model.flows):
        out[f"flow{j}_t"] = model.flow_t[j]
        for key in ("forward", "backward"):
            out[f"flow{j}_{key}"] = f[key].detach()
            out[f"flow{j}_{key}_grad"] = z(f[key].grad, f[key])
    return {f"{tag}__{k}": v for k, v in out.items()}


CASES = (
    # tag, rays, samples, seed, frame, options
    ("default", 96, 24, 1, 2, {}),                                                      # three primary terms + ray chamfer
    ("flow_mid", 96, 24, 2, 2, dict(flow_loss=True)),                                   # both neighbours at both steps
    ("flow_first", 64, 16, 3, 0, dict(flow_loss=True)),                                 # no backward neighbours
    ("flow_last", 64, 16, 4, 4, dict(flow_loss=True)),                                  # no forward neighbours
    ("urf", 64, 48, 5, 1, dict(urf_loss=True)),
    ("crit_huber_bce_l1", 96, 8, 6, 2, dict(depth_loss="huber", raydrop_loss="bce", intensity_loss="l1")),
    ("crit_mse_l1_huber", 96, 8, 7, 2, dict(depth_loss="mse", raydrop_loss="l1", intensity_loss="huber", alpha_d=0.7, alpha_r=0.05, alpha_i=0.2,
                                         smooth_factor=0.1)),
    ("patch_l1", 96, 8, 8, 2, dict(patch_size_lidar=[2, 8], grad_loss=True)),
    ("patch_sobel_cos_all", 128, 8, 9, 2, dict(patch_size_lidar=[4, 8], grad_loss=True, sobel_grad=True, depth_grad_loss="cos",
                                             grad_norm_smooth=True, spatial_smooth=True, tv_loss=True, alpha_grad=0.3)),
    ("patch_mse_tv", 72, 8, 10, 2, dict(patch_size_lidar=3, grad_loss=True, depth_grad_loss="mse", tv_loss=True)),
    ("everything", 96, 32, 11, 2, dict(flow_loss=True, urf_loss=True, patch_size_lidar=[2, 4], grad_loss=True, depth_grad_loss="huber",
                                       spatial_smooth=True)),
)


def main():
    torch.set_num_threads(4)
    Trainer, scratch = import_reference_trainer()
    arrays = {"cases": np.array([c[0] for c in CASES])}
    for tag, n, T, seed, frame, kw in CASES:
        arrays.update(run_case(Trainer, tag, n, T, seed, frame, **kw))
        print(f"{tag}: loss {float(arrays[tag + '__loss']):.6f}")
    save("train_step_losses", **arrays)
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
