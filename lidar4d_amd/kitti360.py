"""Reader of the reference's preprocessed KITTI-360 layout (the on-disk formats next to the path).
Mirror of data/kitti360_dataset.py:14-209 -- same constructor fields, attributes and per-step dict -- so that
``main_lidar4d.py``'s ``KITTI360Dataset(...).dataloader()`` can be pointed here unchanged.

Layout (written by the reference's preprocess scripts):
  ``<root>/transforms_<seq>_<split>.json``  {"h_lidar", "w_lidar", "frames": [{"lidar2world": 4x4, "lidar_file_path",
                                             "frame_id"}, ...]}
  ``<root>/<lidar_file_path>.npy``          range view [H, W, 3] float: (unused, intensity, depth in metres; 0 = no return)

Everything here is host / torch tensor logic (no HIP kernel): frames are parsed once, moved to ``device`` (the GPU's HBM
when ``preload``), and ``collate`` cuts one frame's batch with lidar4d_amd.data.get_lidar_rays.
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from .data import get_lidar_rays

# first and last frame id of the sequences the reference knows (kitti360_dataset.py:29-71)
SEQUENCE_FRAMES = {
    "1538": (1538, 1601), "1728": (1728, 1791), "1908": (1908, 1971), "3353": (3353, 3416),
    "2350": (2350, 2400), "4950": (4950, 5000), "8120": (8120, 8170), "10200": (10200, 10250),
    "10750": (10750, 10800), "11400": (11400, 11450),
}


class KITTI360Dataset:
    def __init__(self, device="cpu", split="train", root_path="data/kitti360", sequence_id="4950", preload=True, scale=1,
                 offset=(), fp16=True, patch_size_lidar=1, num_rays_lidar=4096, fov_lidar=()):
        if str(sequence_id) not in SEQUENCE_FRAMES:
            raise ValueError(f"Invalid sequence id: {sequence_id}")
        self.device, self.root_path, self.sequence_id = device, root_path, str(sequence_id)
        self.preload, self.scale, self.offset, self.fp16 = preload, scale, list(offset), fp16
        self.patch_size_lidar, self.fov_lidar = patch_size_lidar, list(fov_lidar)
        self.frame_start, self.frame_end = SEQUENCE_FRAMES[self.sequence_id]

        # 'refine' reads the training frames but serves whole frames (U-Net refinement, runner.py:818-863)
        self.training = split in ("train", "all", "trainval")
        self.num_rays_lidar = num_rays_lidar if self.training else -1
        self.split = "train" if split == "refine" else split

        with open(os.path.join(root_path, f"transforms_{self.sequence_id}_{self.split}.json")) as fh:
            meta = json.load(fh)
        self.H = int(meta["h"]) if "h" in meta and "w" in meta else None
        self.W = int(meta["w"]) if "h" in meta and "w" in meta else None
        self.H_lidar, self.W_lidar = int(meta["h_lidar"]), int(meta["w_lidar"])

        frames = sorted(meta["frames"], key=lambda fr: fr["lidar_file_path"])
        span = self.frame_end - self.frame_start
        poses = np.stack([np.asarray(fr["lidar2world"], dtype=np.float32) for fr in frames])
        times = np.asarray([(fr["frame_id"] - self.frame_start) / span for fr in frames], dtype=np.float32)
        images = []
        for fr in frames:
            view = np.load(os.path.join(root_path, fr["lidar_file_path"]))       # [H, W, 3]
            depth = view[:, :, 2]
            returned = np.where(depth == 0.0, 0.0, 1.0)
            images.append(np.stack([returned, view[:, :, 1], depth * self.scale], axis=-1))
        # scene normalisation of the sensor positions (configs/kitti360_*.txt: offset, scale); the offset list makes the
        # arithmetic float64 before it is stored back as float32, like the reference's numpy expression
        off = np.asarray(self.offset, dtype=np.float64) if len(self.offset) else np.zeros(3)
        poses[:, :3, 3] = (poses[:, :3, 3] - off) * self.scale
        self.poses_lidar = torch.from_numpy(poses)                               # [N, 4, 4]
        self.images_lidar = torch.from_numpy(np.stack(images)).float()           # [N, H, W, 3]
        self.times = torch.from_numpy(times).view(-1, 1)                         # [N, 1]
        if preload:
            self.poses_lidar = self.poses_lidar.to(device)
            self.images_lidar = self.images_lidar.to(torch.half if fp16 else torch.float).to(device)
            self.times = self.times.to(device)
        self.intrinsics_lidar = self.fov_lidar

    def collate(self, index):
        """index: list with one frame number (the loader's batch size is 1) -> the reference's per-step dict."""
        B = len(index)
        poses = self.poses_lidar[index].to(self.device)
        rays = get_lidar_rays(poses, self.intrinsics_lidar, self.H_lidar, self.W_lidar, self.num_rays_lidar,
                              self.patch_size_lidar)
        images = self.images_lidar[index].to(self.device)
        if self.training:  # ground truth of the drawn pixels only
            C = images.shape[-1]
            images = torch.gather(images.view(B, -1, C), 1, rays["inds"].unsqueeze(-1).expand(-1, -1, C))
        return {"H_lidar": self.H_lidar, "W_lidar": self.W_lidar, "rays_o_lidar": rays["rays_o"],
                "rays_d_lidar": rays["rays_d"], "images_lidar": images, "time": self.times[index].to(self.device),
                "poses_lidar": poses}

    def dataloader(self):
        loader = DataLoader(list(range(len(self))), batch_size=1, collate_fn=self.collate, shuffle=self.training,
                            num_workers=0)
        loader._data = self
        loader.has_gt = self.images_lidar is not None
        return loader

    def __len__(self):
        return len(self.poses_lidar)
