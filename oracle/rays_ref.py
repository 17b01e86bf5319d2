"""Oracle restatement of the reference's LiDAR ray generation.  TEST INFRASTRUCTURE ONLY.

Reference: data/base_dataset.py:15-102 (``get_lidar_rays``), whose direction convention is shared
with utils/convert.py:115-124.  Pinned by ``tests/golden/rays_*.npz`` (made by make_golden.py from
the real function).
"""
import numpy as np
import torch


def lidar_rays(pose, fov_up, fov, H, W, inds=None):
    """rays_o, rays_d [n, 3] (fp32 torch) for pixel indices ``inds`` (row-major j*W+i; all pixels
    when None) of an H x W panorama seen from ``pose`` [4,4] (sensor-to-world)."""
    pose = torch.as_tensor(pose, dtype=torch.float32)
    if inds is None:
        inds = torch.arange(H * W)
    inds = torch.as_tensor(inds, dtype=torch.int64)
    i = (inds % W).float()  # column
    j = (inds // W).float()  # row
    beta = -(i - W / 2) / W * 2 * np.pi
    alpha = (fov_up - j / H * fov) / 180 * np.pi
    d = torch.stack([torch.cos(alpha) * torch.cos(beta), torch.cos(alpha) * torch.sin(beta), torch.sin(alpha)], -1)
    rays_d = d @ pose[:3, :3].t()
    rays_o = pose[:3, 3].expand_as(rays_d)
    return rays_o, rays_d
