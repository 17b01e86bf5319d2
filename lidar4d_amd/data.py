"""LiDAR ray generation and a synthetic KITTI-360-shaped dataset.

``get_lidar_rays`` restates the reference's data/base_dataset.py:15-102 (same direction convention as
utils/convert.py:115-124) on whatever device the pose lives on.  ``SyntheticKitti360`` stands in for
data/kitti360_dataset.py (the KITTI-360 files are not available): same per-step dict keys
(kitti360_dataset.py:177-187), 64 x 1024 panorama, ``fov_lidar = (2.0, 26.9)``, 51 frames on a straight
1 m/frame track, scale/offset convention of configs/kitti360_4950.txt, ground truth from an analytic scene
(ground plane + boxes, 10 % random ray drops); SURVEY.md section 8(d).
"""
import numpy as np
import torch

KITTI360_SCALE = 0.010504329815187737  # configs/kitti360_4950.txt:6
KITTI360_FOV = (2.0, 26.9)


def pixel_block_order(rows, cols):
    """Permutation that puts pixels of the same 8 x 8 block of the range image next to each other (block columns left to
    right, blocks top to bottom inside a column).  Neighbouring pixels are neighbouring rays: their samples touch the
    same plane texels and coarse hash cells, and processed together (and on one XCD: csrc/common.h xcd_tile) those
    table lines are fetched into L2 once.  The result of a training step does not depend on the ray order (the losses
    are sums over rays), so this is a free choice of the data loader.  (Measured: not a win -- see SyntheticKitti360.)"""
    key = ((cols >> 3) << 9) | ((rows >> 3) << 6) | ((cols & 7) << 3) | (rows & 7)
    return torch.argsort(key)


def get_lidar_rays(poses, intrinsics, H, W, N=-1, patch_size=1, generator=None, sort_pixels=False):
    """poses [B,4,4] sensor-to-world, intrinsics (fov_up, fov) in degrees -> dict(rays_o, rays_d [B,n,3], inds [B,n]).
    N > 0 draws pixels the way the reference does (base_dataset.py:36-70): ``N // (px * py)`` patches of px x py pixels
    (``patch_size`` an int or [px, py]) with the top row in [0, H - px) and the left column in [0, W), columns wrapping
    around the panorama; px <= 0 draws N pixels anywhere, possibly repeated.  With the default patch_size = 1 that is
    a row in [0, H-1) -- the last row is never sampled -- and a column in [0, W).  Same torch RNG consumption as the
    reference, so equal seeds give equal pixels.  sort_pixels: serve the drawn pixels in pixel_block_order."""
    device = poses.device
    B = poses.shape[0]
    if N > 0:
        N = min(N, H * W)
        if isinstance(patch_size, int):
            px, py = patch_size, patch_size
        elif len(patch_size) == 1:
            px, py = patch_size[0], patch_size[0]
        else:
            px, py = patch_size
        if px > 0:
            n_patch = N // (px * py)
            top = torch.randint(0, H - px, size=[n_patch], device=device, generator=generator)
            left = torch.randint(0, W, size=[n_patch], device=device, generator=generator)
            dr = torch.arange(px, device=device).repeat_interleave(py)  # row offsets, patch-row major
            dc = torch.arange(py, device=device).repeat(px)             # column offsets
            rows = (top[:, None] + dr[None, :]).reshape(-1)
            cols = ((left[:, None] + dc[None, :]) % W).reshape(-1)
        else:
            flat = torch.randint(0, H * W, size=[N], device=device, generator=generator)
            rows, cols = torch.div(flat, W, rounding_mode="floor"), flat % W
        if rows.numel() != N:
            raise ValueError(f"get_lidar_rays: N = {N} is not a multiple of the patch size {px} x {py}")
        if sort_pixels:
            order = pixel_block_order(rows, cols)
            rows, cols = rows[order], cols[order]
        inds = (rows * W + cols).expand([B, N])
    else:
        inds = torch.arange(H * W, device=device).expand([B, H * W])
    i = (inds % W).float()
    j = torch.div(inds, W, rounding_mode="floor").float()
    fov_up, fov = intrinsics
    beta = -(i - W / 2) / W * 2 * np.pi
    alpha = (fov_up - j / H * fov) / 180 * np.pi
    directions = torch.stack([torch.cos(alpha) * torch.cos(beta), torch.cos(alpha) * torch.sin(beta), torch.sin(alpha)], -1)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds}


def _analytic_scene_depth(origin, dirs):
    """Metric range along unit ``dirs`` [n,3] from ``origin`` [3] to a ground plane z = -1.7 and five boxes."""
    inf = torch.full(dirs.shape[:1], float("inf"), device=dirs.device)
    dz = dirs[:, 2]
    t_ground = torch.where(dz < -1e-6, (-1.7 - origin[2]) / dz, inf)
    best = t_ground
    boxes = [((8, -3, -1.7), (12, 1, 0.3)), ((-20, 6, -1.7), (-14, 10, 2.5)), ((15, 12, -1.7), (40, 14, 6.0)),
             ((-5, -16, -1.7), (30, -14, 4.0)), ((30, -4, -1.7), (34, 0, 0.0))]
    for lo, hi in boxes:
        lo = torch.tensor(lo, dtype=torch.float32, device=dirs.device)
        hi = torch.tensor(hi, dtype=torch.float32, device=dirs.device)
        inv = 1.0 / torch.where(dirs.abs() < 1e-9, torch.full_like(dirs, 1e-9), dirs)
        t0, t1 = (lo - origin) * inv, (hi - origin) * inv
        tmin = torch.minimum(t0, t1).amax(-1)
        tmax = torch.maximum(t0, t1).amin(-1)
        hit = (tmax >= tmin) & (tmin > 0)
        best = torch.minimum(best, torch.where(hit, tmin, inf))
    return best


class SyntheticKitti360:
    """Pre-loads ``num_frames`` synthetic range images [H,W,3] = (ray-drop mask, intensity, depth*scale) on ``device``
    and serves per-step ray batches like KITTI360Dataset.collate (one frame per step)."""

    def __init__(self, device, H=64, W=1024, num_frames=51, num_rays=4096, scale=KITTI360_SCALE, fov=KITTI360_FOV, seed=0,
                 sort_pixels=False, frame_seed=None):
        self.device, self.H, self.W, self.num_frames, self.num_rays = device, H, W, num_frames, num_rays
        self.patch_size_lidar = 1  # settable like the reference's dataset attribute (runner.py:700-705): int or [px, py]
        self.fused_batch = True  # batch_for: one HIP launch behind the two random draws (False: the torch restatement)
        self.sort_pixels = sort_pixels  # pixel_block_order; measured SLOWER on MI355X (66.3 vs 63.5 ms/step: neighbouring rays
        # pile onto the same LDS histogram bins / cache lines), kept as an option for experiments
        self.scale, self.fov = scale, fov
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        # host-side frame choice: no device sync.  frame_seed: data-parallel ranks that share it step through the same
        # frame sequence (each with its own rays) -- the per-step work depends on the frame, so this keeps ranks in step
        self.frame_gen = torch.Generator().manual_seed(seed if frame_seed is None else frame_seed)
        g_cpu = torch.Generator().manual_seed(1234)
        self.poses, self.images = [], []
        for k in range(num_frames):
            pose_m = torch.eye(4)
            pose_m[0, 3] = float(k - num_frames // 2)  # 1 m per frame along x, centred on the scene offset
            rays = get_lidar_rays(pose_m[None], fov, H, W, -1)
            depth = _analytic_scene_depth(pose_m[:3, 3], rays["rays_d"][0])
            valid = (depth < 80.0) & (torch.rand(H * W, generator=g_cpu) > 0.1)
            depth = torch.where(valid, depth, torch.zeros_like(depth))
            hit = pose_m[:3, 3] + rays["rays_d"][0] * depth[:, None]
            intensity = (0.5 + 0.5 * torch.sin(0.7 * hit[:, 0]) * torch.cos(0.5 * hit[:, 1])).clamp(0, 1) * valid
            img = torch.stack([valid.float(), intensity, depth * scale], -1).view(H, W, 3)
            pose = pose_m.clone()
            pose[:3, 3] = pose[:3, 3] * scale
            self.poses.append(pose.to(device))
            self.images.append(img.to(device))
        self.poses = torch.stack(self.poses)
        self.images = torch.stack(self.images)
        # frame times resident on the device: building the [1, 1] tensor per batch is a pageable host-to-device copy, which
        # a captured training step (hipGraph, Trainer.train_step_graphed) cannot contain
        self.times = torch.tensor([[k / (num_frames - 1)] for k in range(num_frames)], dtype=torch.float32, device=device).view(num_frames, 1, 1)

    def next_frame(self):
        """The frame of the next training step (host-side draw: no device sync)."""
        return int(torch.randint(0, self.num_frames, [1], generator=self.frame_gen))

    def batch(self, frame=None):
        """Dict with the reference's keys for one training step (random frame unless given)."""
        if frame is None:
            frame = self.next_frame()
        return self.batch_for(frame)

    def frame(self, frame, W=None):
        """Every ray of one frame, as the evaluation / simulation loops feed ``render(staged=True)``
        (kitti360_dataset.py:150-180 with num_rays = -1).  ``W`` overrides the image width (novel-view renders use
        2048 columns, BASELINE config 5); ground truth is only attached at the dataset's own width."""
        W = W or self.W
        pose = self.poses[frame:frame + 1]
        rays = get_lidar_rays(pose, self.fov, self.H, W, -1)
        t = torch.tensor([[frame / (self.num_frames - 1)]], dtype=torch.float32, device=self.device)
        out = {"rays_o_lidar": rays["rays_o"], "rays_d_lidar": rays["rays_d"], "time": t, "poses_lidar": pose,
               "H_lidar": self.H, "W_lidar": W, "index": [frame]}
        if W == self.W:
            out["images_lidar"] = self.images[frame:frame + 1]
        return out

    def batch_for(self, frame):
        pose = self.poses[frame:frame + 1]
        t = self.times[frame]
        # (num_rays <= 0 = "every pixel of the frame" in get_lidar_rays, kitti360_dataset.py:150-180: that case takes the torch path)
        if (self.fused_batch and self.num_rays > 0 and torch.device(self.device).type == "cuda" and self.patch_size_lidar == 1
                and not self.sort_pixels):
            # the two draws of get_lidar_rays (same generator consumption), then ONE launch for pixel index, direction,
            # rotation, origin and the ground-truth gather (csrc/glue.hip: 33 torch launches otherwise)
            from . import ops
            n = min(self.num_rays, self.H * self.W)
            top = torch.randint(0, self.H - 1, size=[n], device=self.device, generator=self.gen)
            left = torch.randint(0, self.W, size=[n], device=self.device, generator=self.gen)
            rays_o, rays_d, images, _ = ops.lidar_ray_batch(top, left, self.poses[frame], self.fov, self.H, self.W, self.images[frame])
            return {"rays_o_lidar": rays_o, "rays_d_lidar": rays_d, "time": t, "images_lidar": images,
                    "poses_lidar": pose, "H_lidar": self.H, "W_lidar": self.W, "index": [frame],
                    "time_host": frame / (self.num_frames - 1)}
        rays = get_lidar_rays(pose, self.fov, self.H, self.W, self.num_rays, self.patch_size_lidar, generator=self.gen,
                              sort_pixels=self.sort_pixels and self.patch_size_lidar == 1)
        inds = rays["inds"]
        images = torch.gather(self.images[frame].view(1, -1, 3), 1, inds[..., None].expand(-1, -1, 3))
        return {"rays_o_lidar": rays["rays_o"], "rays_d_lidar": rays["rays_d"], "time": t, "images_lidar": images,
                "poses_lidar": pose, "H_lidar": self.H, "W_lidar": self.W, "index": [frame],
                "time_host": frame / (self.num_frames - 1)}  # the same number on the host: no read-back for host-side decisions
