"""Dataset preparation either side of the on-disk format the path trains on.  Mirrors of the
reference's data/preprocess/generate_rangeview.py:28-70 (raw Velodyne scans -> range views) and
data/preprocess/cal_seq_config.py:28-69,92-111 (scene scale / offset of a sequence -> configs/<dataset>_<seq>.txt), with
the point <-> range-image conversions on the device (lidar4d_amd.convert; the reference walks every point of every scan
in a python loop).

The conversion functions are parameters (``to_pano`` / ``to_points``) so that the file handling and the reductions can be
exercised without a GPU; they default to the HIP kernels.
"""
import os

import numpy as np
import torch


def range_view_from_points(points, H, W, intrinsics, max_depth=80.0, to_pano=None):
    """generate_rangeview.py:28-42 (``LiDAR_2_Pano_KITTI``): points [n, 4] (x, y, z, intensity), sensor frame, metres ->
    range view [H, W, 3] = (0, intensity, range)."""
    if to_pano is None:
        from .convert import lidar_to_pano_with_intensities as to_pano
    pano, intensities = to_pano(points, H, W, intrinsics, max_depth)
    pano, intensities = torch.as_tensor(pano), torch.as_tensor(intensities)
    view = torch.zeros(H, W, 3, dtype=torch.float64, device=pano.device)  # the reference's np.zeros: float64 files
    view[:, :, 1] = intensities
    view[:, :, 2] = pano
    return view


def generate_rangeview(lidar_paths, out_dir, H=66, W=1030, intrinsics=(2.0, 26.9), points_dim=4, device="cuda", to_pano=None):
    """generate_rangeview.py:45-70: one ``<frame>.npy`` range view per raw ``<frame>.bin`` scan (float32 records of
    ``points_dim`` values).  Returns the written paths."""
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for path in lidar_paths:
        scan = np.fromfile(path, dtype=np.float32).reshape(-1, points_dim)
        pts = torch.from_numpy(scan[:, :4].copy()).to(device)
        view = range_view_from_points(pts, H, W, intrinsics, to_pano=to_pano)
        name = os.path.splitext(os.path.basename(path))[0] + ".npy"
        np.save(os.path.join(out_dir, name), view.cpu().numpy())
        written.append(os.path.join(out_dir, name))
    return written


def cal_centerpose_bound_scale(lidar_rangeview_paths, lidar2worlds, fov_lidar, bound=1.0, device="cuda", to_points=None):
    """cal_seq_config.py:28-69: world-space bounding box of all returns of a sequence -> (scale, centerpose) such that
    the centred scene fits [-bound, bound]^3 ((max + min) / 2 per axis; scale = bound / largest positive extent), plus
    the nearest / farthest range.  Reductions in float64 like the reference's numpy; nothing is concatenated (running
    minima / maxima per frame)."""
    if to_points is None:
        from .convert import pano_to_lidar as to_points
    near, far = 200.0, 0.0
    lo = torch.full((3,), float("inf"), dtype=torch.float64)
    hi = torch.full((3,), float("-inf"), dtype=torch.float64)
    for path, l2w in zip(lidar_rangeview_paths, lidar2worlds):
        pano = np.load(path)
        pts = torch.as_tensor(to_points(torch.from_numpy(np.ascontiguousarray(pano[:, :, 2])).to(device), fov_lidar)).double()
        if pts.shape[0] == 0:
            continue
        dis = torch.sqrt(pts[:, 0] ** 2 + pts[:, 1] ** 2 + pts[:, 2] ** 2)
        near, far = min(near, float(dis.min())), max(far, float(dis.max()))
        homo = torch.cat([pts, torch.ones(pts.shape[0], 1, dtype=torch.float64, device=pts.device)], -1)
        world = (homo @ torch.as_tensor(np.asarray(l2w), dtype=torch.float64, device=pts.device).T)[:, :3]
        lo = torch.minimum(lo, world.min(0).values.cpu())
        hi = torch.maximum(hi, world.max(0).values.cpu())
    centerpose = [float((hi[k] + lo[k]) / 2.0) for k in range(3)]
    bound_ori = [float(hi[k]) - centerpose[k] for k in range(3)]
    scale = bound / max(bound_ori)
    return scale, centerpose, near, far


def write_seq_config(config_path, dataset, root_path, sequence_id, num_frames, fov_lidar, scale, centerpose):
    """cal_seq_config.py:102-111: the per-sequence config file main_lidar4d.py reads (configs/kitti360_4950.txt)."""
    entries = {"dataloader": dataset, "path": root_path, "sequence_id": sequence_id, "num_frames": num_frames,
               "fov_lidar": list(fov_lidar), "scale": scale, "offset": list(centerpose)}
    os.makedirs(os.path.dirname(os.path.abspath(config_path)), exist_ok=True)
    with open(config_path, "w") as fh:
        fh.write("".join(f"{key} = {value}\n" for key, value in entries.items()))
    return config_path
