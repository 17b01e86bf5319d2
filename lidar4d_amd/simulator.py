"""Novel-view LiDAR simulation on top of the render path.  Mirror of the reference's model/simulator.py:22-232 and of the
trajectory edit in main_lidar4d_sim.py:249-275: for every requested sensor pose render the full panorama in chunks
(``render(staged=True)``), refine the ray-drop probability with the U-Net, mask, and turn the range image into a point
cloud.  Everything up to the file output stays on the device (lidar4d_amd.convert instead of the numpy round trip of
simulator.py:137-142).

Outputs: ``<workspace>/points/lidar4d_%04d.npy`` ([n, 4] x, y, z, intensity in metres, sensor frame -- the reference's
format) and, when Pillow is importable, ``<workspace>/images/lidar4d_%04d.png`` (ray-drop | intensity | depth stacked,
grey levels; the reference colours them with OpenCV colour maps and also writes an mp4 through imageio -- neither package
is a dependency here, so the video is skipped and says so).
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F


def shift_trajectory(rays_o, shift_x=0.0, shift_y=0.0, shift_z=0.0, scale=1.0, align_axis=False):
    """main_lidar4d_sim.py:249-275: move every frame's sensor origin by (shift_x, shift_y, shift_z) metres; with
    ``align_axis`` the x / y shifts are taken along / across the vehicle's direction of motion (the direction to the next
    frame's origin; the last frame reuses the previous one).  rays_o: [frames, rays, 3] in scene units."""
    out = rays_o.clone()
    forward = torch.tensor([[1.0, 0.0, 0.0]]).to(rays_o)
    dx, dy = shift_x, shift_y
    n = rays_o.shape[0]
    for i in range(n):
        if align_axis:
            if i < n - 1:
                forward = F.normalize((rays_o[i + 1, 0, :] - rays_o[i, 0, :]).unsqueeze(0), p=2)
            left = torch.stack([-forward[0, 1], forward[0, 0], forward[0, 2]]).unsqueeze(0)
            move = shift_x * forward + shift_y * left
            dx, dy = move[0, 0], move[0, 1]
        out[i, :, 0] += dx * scale
        out[i, :, 1] += dy * scale
        out[i, :, 2] += shift_z * scale
    return out


def _grey(img):
    return (np.clip(img, 0.0, 1.0) * 255).astype(np.uint8)


class Simulator:
    def __init__(self, name, opt, model, device=None, mute=False, fp16=False, workspace="simulation",
                 use_checkpoint="latest_model", use_refine=True, H_lidar=66, W_lidar=1030, to_points=None):
        """opt: namespace with at least ``scale`` and ``fov_lidar`` (every attribute is also forwarded to
        ``model.render`` as keyword, like the reference does).  to_points(depth_m [H,W], intensity [H,W], fov) -> [n,4]
        defaults to the device kernel lidar4d_amd.convert.pano_to_lidar_with_intensities."""
        self.name, self.opt, self.mute, self.fp16 = name, opt, mute, fp16
        self.workspace, self.use_checkpoint, self.use_refine = workspace, use_checkpoint, use_refine
        self.H_lidar, self.W_lidar = H_lidar, W_lidar
        self.time_stamp = time.strftime("%Y-%m-%d_%H-%M-%S")
        self.device = device if device is not None else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
        self.model = model.to(self.device) if hasattr(model, "to") else model
        if to_points is None:
            from .convert import pano_to_lidar_with_intensities as to_points
        self.to_points = to_points
        self.log_ptr = None
        if workspace is not None:
            os.makedirs(workspace, exist_ok=True)
            self.log_ptr = open(os.path.join(workspace, f"log_{name}.txt"), "a+")
            self.ckpt_path = os.path.join(workspace, "checkpoints")
            self.best_path = os.path.join(self.ckpt_path, f"{name}.pth")
        self.log(f"[INFO] Simulator: {name} | {self.time_stamp} | {self.device} | {workspace}")
        if workspace is not None and use_checkpoint != "scratch":
            self._load(use_checkpoint)

    def __del__(self):
        if getattr(self, "log_ptr", None):
            self.log_ptr.close()

    def log(self, *args):
        if not self.mute:
            print(*args)
        if self.log_ptr:
            print(*args, file=self.log_ptr)
            self.log_ptr.flush()

    def _load(self, which):
        """simulator.py:72-92,197-232: 'latest' / 'latest_model' / 'best' / a path.  Only the model weights matter here."""
        from .checkpoint import latest_checkpoint, load_checkpoint
        path = which
        if which in ("latest", "latest_model"):
            path = latest_checkpoint(self.ckpt_path, self.name)
        elif which == "best":
            path = self.best_path if os.path.exists(self.best_path) else latest_checkpoint(self.ckpt_path, self.name)
        if path is None or not os.path.exists(path):
            self.log("[WARN] No checkpoint found, model randomly initialized.")
            return
        info = load_checkpoint(path, self.model, model_only=True, map_location=self.device)
        self.log(f"[INFO] loaded model from {path}" + (f" (missing keys: {info['missing_keys']})" if info["missing_keys"] else ""))

    @torch.no_grad()
    def render(self, rays_o_lidar, rays_d_lidar, times_lidar, save_pc=True, save_img=True, save_video=True):
        """rays_o_lidar / rays_d_lidar: [frames, H*W, 3], times_lidar: [frames, 1].  Returns the last frame's points."""
        H, W = self.H_lidar, self.W_lidar
        opt_kw = {k: v for k, v in vars(self.opt).items()}
        pred_lidar = None
        for i in range(rays_o_lidar.shape[0]):
            out = self.model.render(rays_o_lidar[i:i + 1], rays_d_lidar[i:i + 1], times_lidar[i:i + 1], staged=True, perturb=False,
                                    **opt_kw)
            image = out["image_lidar"].reshape(-1, H, W, 2)
            raydrop, intensity = image[..., 0], image[..., 1]
            depth = out["depth_lidar"].reshape(-1, H, W)
            if self.use_refine:
                raydrop = self.model.unet(torch.cat([raydrop, intensity, depth], dim=0).unsqueeze(0)).squeeze(0)
            keep = torch.where(raydrop > 0.5, 1, 0)
            intensity, depth = intensity * keep, depth * keep
            pred_lidar = self.to_points(depth[0] / self.opt.scale, intensity[0], self.opt.fov_lidar)
            if save_pc and self.workspace is not None:
                path = os.path.join(self.workspace, "points", f"lidar4d_{i:04d}.npy")
                os.makedirs(os.path.dirname(path), exist_ok=True)
                np.save(path, pred_lidar.detach().cpu().numpy() if torch.is_tensor(pred_lidar) else np.asarray(pred_lidar))
            if save_img and self.workspace is not None:
                self._save_image(i, raydrop[0], intensity[0], depth[0])
        if save_video:
            self.log("[INFO] video output needs imageio, which is not a dependency of this build: skipped")
        return pred_lidar

    def _save_image(self, i, raydrop, intensity, depth):
        try:
            from PIL import Image
        except ImportError:
            self.log("[INFO] Pillow not importable: image output skipped")
            return
        rows = [_grey(t.detach().float().cpu().numpy()) for t in (raydrop, intensity, depth)]
        path = os.path.join(self.workspace, "images", f"lidar4d_{i:04d}.png")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(np.concatenate(rows, axis=0)).save(path)
