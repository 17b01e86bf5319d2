"""Dataset preparation either side of the on-disk format the path trains on.  Mirrors of the
reference's data/preprocess/generate_rangeview.py:28-70 (raw Velodyne scans -> range views),
kitti360_loader.py:62-127 + kitti360_to_nerf.py:76-146 (raw poses and calibration -> transforms_<seq>_<split>.json) and
cal_seq_config.py:28-69,92-111 (scene scale / offset of a sequence -> configs/<dataset>_<seq>.txt), with
the point <-> range-image conversions on the device (lidar4d_amd.convert; the reference walks every point of every scan
in a python loop).

The conversion functions are parameters (``to_pano`` / ``to_points``) so that the file handling and the reductions can be
exercised without a GPU; they default to the HIP kernels.
"""
import json
import os

import numpy as np
import torch

from .kitti360 import SEQUENCE_FRAMES

# held-out frames of each sequence (kitti360_to_nerf.py:32-73): four frames, every 13th of the 64-frame sequences and
# every 10th of the 51-frame ones; "test" is the same set as "val"
VAL_FRAMES = {seq: [lo + (13 if hi - lo == 63 else 10) * k for k in range(1, 5)] for seq, (lo, hi) in SEQUENCE_FRAMES.items()}


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


def load_lidar_poses(kitti_360_root, sequence_name, frame_ids):
    """data/preprocess/kitti360_loader.py:62-127: sensor-to-world matrices of the Velodyne from KITTI-360's raw files --
    ``data_poses/<seq>_sync/poses.txt`` (frame id + 3x4 IMU-to-world), ``calibration/calib_cam_to_pose.txt`` (``image_00:``
    3x4 camera-to-IMU) and ``calibration/calib_cam_to_velo.txt`` (3x4 camera-to-Velodyne):
    velo_to_world = imu_to_world @ cam00_to_imu @ inv(cam00_to_velo); frames without a pose reuse the previous one.
    (Restated from the reference, which needs the ``camtools`` package that is not available here: unpinned.)"""
    pad = lambda m: np.vstack([np.asarray(m, dtype=np.float64).reshape(3, 4), [0.0, 0.0, 0.0, 1.0]])
    imu_to_world = {}
    for row in np.loadtxt(os.path.join(kitti_360_root, "data_poses", f"{sequence_name}_sync", "poses.txt"), ndmin=2):
        imu_to_world[int(row[0])] = row[1:].reshape(3, 4)
    cam_to_imu = None
    with open(os.path.join(kitti_360_root, "calibration", "calib_cam_to_pose.txt")) as fh:
        for line in fh:
            if line.startswith("image_00"):
                cam_to_imu = pad([float(v) for v in line.split(":", 1)[1].split()])
    if cam_to_imu is None:
        raise ValueError("calib_cam_to_pose.txt has no image_00 entry")
    with open(os.path.join(kitti_360_root, "calibration", "calib_cam_to_velo.txt")) as fh:
        cam_to_velo = pad([float(v) for v in fh.readline().split()])
    velo_to_cam = np.linalg.inv(cam_to_velo)
    out, last = [], None
    for fid in frame_ids:
        if fid in imu_to_world:
            last = pad(imu_to_world[fid] @ cam_to_imu @ velo_to_cam)
        if last is None:
            raise ValueError(f"no pose at or before frame {fid}")
        out.append(last)
    return np.stack(out)


def write_transforms(root, sequence_id, lidar2world, range_view_dir="train"):
    """data/preprocess/kitti360_to_nerf.py:76-146: the ``transforms_<seq>_{train,val,test}.json`` files the dataset reader
    loads -- lidar2world [frames, 4, 4] for every frame of the sequence (first to last id inclusive), range views expected
    as ``<root>/<range_view_dir>/<frame id, 10 digits>.npy``.  Returns the three paths."""
    sequence_id = str(sequence_id)
    first, last = SEQUENCE_FRAMES[sequence_id]
    frame_ids = list(range(first, last + 1))
    if len(lidar2world) != len(frame_ids):
        raise ValueError(f"sequence {sequence_id} has {len(frame_ids)} frames, got {len(lidar2world)} poses")
    h, w, _ = np.load(os.path.join(root, range_view_dir, f"{first:010d}.npy")).shape
    held_out = VAL_FRAMES[sequence_id]
    splits = {"train": [f for f in frame_ids if f not in held_out], "val": held_out, "test": held_out}
    paths = []
    for split, ids in splits.items():
        doc = {"w_lidar": int(w), "h_lidar": int(h), "num_frames": len(frame_ids), "num_frames_split": len(ids),
               "frames": [{"frame_id": f, "lidar_file_path": os.path.join(range_view_dir, f"{f:010d}.npy"),
                           "lidar2world": np.asarray(lidar2world[f - first]).tolist()} for f in ids]}
        paths.append(os.path.join(root, f"transforms_{sequence_id}_{split}.json"))
        with open(paths[-1], "w") as fh:
            json.dump(doc, fh, indent=2)
    return paths
