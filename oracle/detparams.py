"""Deterministic, RNG-library-independent parameter and input fills.  TEST INFRASTRUCTURE ONLY.

Full-size hash tables are far too large to commit as fixtures, so golden generation
(``make_golden.py``, which fills the *reference's* modules) and the tests (which fill the oracle's
and the HIP path's modules) regenerate identical values from an integer hash of the element index
and a per-tensor key.  Pure numpy uint64 arithmetic: independent of torch's RNG implementation.
"""
import zlib

import numpy as np
import torch


def unit_hash(n, key, offset=0):
    """n floats in [0, 1), a function of (key, offset + i) only (splitmix64 finaliser)."""
    i = np.arange(offset, offset + n, dtype=np.uint64)
    k = np.uint64(zlib.crc32(key.encode()) if isinstance(key, str) else int(key))
    with np.errstate(over="ignore"):
        z = i * np.uint64(0x9E3779B97F4A7C15) + (k + np.uint64(1)) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def det_uniform(shape, key, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit_hash(n, key)
    return torch.from_numpy((np.float32(lo) + np.float32(hi - lo) * u).astype(np.float32)).reshape(shape)


def fill_model(model, seed=0, hash_amp=0.5, flow_out_amp=0.02):
    """Fill every parameter of a LiDAR4D-shaped module tree (reference, oracle or HIP build: they
    share state-dict keys) with values that make every branch of the path numerically visible:
    hash tables U(-hash_amp, hash_amp); static planes U(0.1, 0.5); time planes U(0.8, 1.2);
    MLP weight matrices U(-b, b) with b = sqrt(6 / (64 + 64)); flow output layer U(-a, a)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.numel() == 0 or name.startswith("unet."):
                continue
            key = f"{seed}:{name}"
            if name.startswith("planes_encoder."):
                is_time = name.split(".")[-1] in ("2", "4", "5")  # combs (0,3) (1,3) (2,3)
                lo, hi = (0.8, 1.2) if is_time else (0.1, 0.5)
            elif name.endswith("mlp.4.weight") or (name.startswith("flow_net.mlp") and p.shape[0] == 6):
                lo, hi = -flow_out_amp, flow_out_amp
            elif name.startswith("flow_net.mlp") or name.endswith("_net.params"):
                b = float(np.sqrt(6.0 / 128.0))
                lo, hi = -b, b
            else:  # hash tables
                lo, hi = -hash_amp, hash_amp
            p.copy_(det_uniform(tuple(p.shape), key, lo, hi).to(p.dtype))
    return model


def grad_digest(model):
    """Per-parameter (sum, sum|.|, dot with a deterministic probe vector) of ``.grad``: a compact,
    order-independent-enough fingerprint of gradients too large to store."""
    out = {}
    for name, p in model.named_parameters():
        if p.numel() == 0 or name.startswith("unet."):
            continue
        g = p.grad
        if g is None:
            out[name] = np.zeros(3, dtype=np.float64)
            continue
        g = g.detach().double().reshape(-1).cpu()
        probe = torch.from_numpy(unit_hash(g.numel(), "probe:" + name)).double() - 0.5
        out[name] = np.array([g.sum().item(), g.abs().sum().item(), (g * probe).sum().item()])
    return out


def fill_unet(unet, seed=0):
    """Deterministic fill of a U-Net shaped module tree (reference model/unet.py or lidar4d_amd/unet.py: same
    state-dict keys): conv weights U(-b, b) with b = sqrt(3 / fan_in), conv biases U(-0.1, 0.1), batch-norm scale
    U(0.5, 1.5), shift and running mean U(-0.2, 0.2), running variance U(0.5, 1.5)."""
    with torch.no_grad():
        for name, t in list(unet.named_parameters()) + list(unet.named_buffers()):
            key = f"unet{seed}:{name}"
            leaf = name.split(".")[-1]
            if leaf == "num_batches_tracked":
                continue
            if t.dim() == 4:
                b = float(np.sqrt(3.0 / (t.shape[1] * t.shape[2] * t.shape[3])))
                lo, hi = -b, b
            elif leaf == "running_var" or (leaf == "weight" and t.dim() == 1):
                lo, hi = 0.5, 1.5
            elif leaf == "running_mean" or leaf == "bias":
                lo, hi = (-0.2, 0.2) if leaf == "running_mean" or "conv" not in name.split(".")[-2:] else (-0.1, 0.1)
            else:
                raise KeyError(name)
            t.copy_(det_uniform(tuple(t.shape), key, lo, hi).to(t.dtype))
    return unet


def convert_inputs(H=64, W=1024):
    """(depth [H,W], intensity [H,W], cloud [6000,4]) float32 numpy: the inputs of the conversion fixtures."""
    depth = det_uniform((H, W), "cv_depth", 1.5, 79.0).numpy()
    depth[unit_hash(H * W, "cv_drop").reshape(H, W) < 0.1] = 0.0  # 10 % dropped rays
    inten = det_uniform((H, W), "cv_int", 0.0, 1.0).numpy()
    # a cloud that does not sit on pixel centres, with duplicates per pixel, far points and out-of-fov points
    n = 6000
    az = det_uniform((n,), "cv_az", -np.pi, np.pi).numpy()
    el = det_uniform((n,), "cv_el", np.deg2rad(-30.0), np.deg2rad(6.0)).numpy()
    rng = det_uniform((n,), "cv_r", 0.5, 95.0).numpy()
    cloud = np.stack([rng * np.cos(el) * np.cos(az), rng * np.cos(el) * np.sin(az), rng * np.sin(el),
                      det_uniform((n,), "cv_i", 0.0, 1.0).numpy()], -1).astype(np.float32)
    cloud[100:200] = cloud[0:100]            # exact duplicates (first wins)
    cloud[200:300, :3] = cloud[0:100, :3]    # same position, other intensity
    return depth, inten, cloud


def write_kitti360_fixture(root, H=8, W=32, n_train=4, n_val=2):
    """A tiny sequence in the reference's preprocessed KITTI-360 layout (transforms_<seq>_<split>.json + range-view
    .npy files) with deterministic contents; returns the dataset arguments that go with it."""
    import json
    import os

    cfg = {"sequence_id": "4950", "scale": 0.010504329815187737, "offset": [1012.5, 3792.25, 115.75], "fov_lidar": [2.0, 26.9]}
    start = 4950
    ids = {"train": [start + 1 + 3 * k for k in range(n_train)], "val": [start + 2 + 5 * k for k in range(n_val)]}
    for split, frame_ids in ids.items():
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for fid in reversed(frame_ids):  # unsorted on purpose: the reader sorts by file path
            ang = float(det_uniform((1,), f"k360a{fid}", -0.5, 0.5))
            c, s = float(np.cos(ang)), float(np.sin(ang))
            pos = det_uniform((3,), f"k360p{fid}", -20.0, 20.0).numpy() + np.asarray(cfg["offset"], dtype=np.float32)
            pose = [[c, -s, 0.0, float(pos[0])], [s, c, 0.0, float(pos[1])], [0.0, 0.0, 1.0, float(pos[2])], [0.0, 0.0, 0.0, 1.0]]
            view = np.zeros((H, W, 3), dtype=np.float32)
            view[:, :, 1] = det_uniform((H, W), f"k360i{fid}", 0.0, 1.0).numpy()
            depth = det_uniform((H, W), f"k360d{fid}", 2.0, 78.0).numpy()
            depth[unit_hash(H * W, f"k360m{fid}").reshape(H, W) < 0.15] = 0.0
            view[:, :, 2] = depth
            rel = os.path.join(split, f"{fid:010d}.npy")
            np.save(os.path.join(root, rel), view)
            frames.append({"frame_id": fid, "lidar_file_path": rel, "lidar2world": pose})
        with open(os.path.join(root, f"transforms_{cfg['sequence_id']}_{split}.json"), "w") as fh:
            json.dump({"h_lidar": H, "w_lidar": W, "frames": frames}, fh)
    return cfg


def write_scan_fixture(root, n_frames=3, n_points=4000):
    """Raw-scan side of the preprocessing fixtures: ``<frame>.bin`` files (float32 x, y, z, intensity like KITTI-360's
    velodyne_points) with deterministic contents, and a sensor-to-world pose per frame.  Returns (bin paths, poses)."""
    import os

    paths, poses = [], []
    for k in range(n_frames):
        az = det_uniform((n_points,), f"pp_az{k}", -np.pi, np.pi).numpy()
        el = det_uniform((n_points,), f"pp_el{k}", np.deg2rad(-24.0), np.deg2rad(1.5)).numpy()
        r = det_uniform((n_points,), f"pp_r{k}", 1.0, 90.0).numpy()
        cloud = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el),
                          det_uniform((n_points,), f"pp_i{k}", 0, 1).numpy()], -1).astype(np.float32)
        path = os.path.join(root, f"{k:010d}.bin")
        cloud.tofile(path)
        paths.append(path)
        a = 0.3 * k
        c, s = np.cos(a), np.sin(a)
        poses.append(np.array([[c, -s, 0, 100 + 5 * k], [s, c, 0, -40 + 2 * k], [0, 0, 1, 3 + 0.1 * k], [0, 0, 0, 1]], dtype=np.float32))
    return paths, poses
