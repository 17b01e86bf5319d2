"""Range image <-> point cloud conversions.  Mirror of the reference's utils/convert.py:4-156 (same function names,
argument order and meaning) on the HIP kernels of lidar4d_amd/csrc/convert.hip, so that evaluation
(utils/metrics.py:253-254), point-cloud export (model/runner.py:764-767) and the simulator (model/simulator.py:137-142)
stay on the device instead of round-tripping every rendered frame through numpy and a python loop.

Inputs are HIP tensors; outputs are HIP tensors (float32).  ``pano_to_lidar*`` needs the number of valid pixels on the
host to size its result, which is the one synchronisation of these functions (pass ``return_count=True`` to get the
padded ``[H*W, 4]`` buffer and the device-side count instead, no sync).
"""
import torch

from . import _lib, ops


def _f32(t, name):
    t = torch.as_tensor(t) if not torch.is_tensor(t) else t
    t = t.detach()
    if not t.is_cuda:
        ops._chk(t, None, name)  # raises: no CPU path
    return t.to(torch.float32).contiguous()


def pano_to_lidar_with_intensities(pano, intensities, lidar_K, return_count=False):
    """convert.py:99-137: pano [H,W], intensities [H,W] (or None), lidar_K = (fov_up, fov) -> [N,4] points."""
    pano = _f32(pano, "pano")
    H, W = pano.shape
    inten = None if intensities is None else _f32(intensities, "intensities").reshape(H, W)
    fov_up, fov = float(lidar_K[0]), float(lidar_K[1])
    dev = pano.device
    pts = torch.empty(H * W, 4, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(1, _lib.lib().l4d_pano_to_lidar_workspace(H, W)), dtype=torch.uint8, device=dev)
    ops.call("l4d_pano_to_lidar", ops._p(pano), ops._p(inten), H, W, fov_up, fov, ops._p(pts), ops._p(count), ops._p(ws),
             ops._stream())
    if return_count:
        return pts, count
    return pts[: int(count.item())]


def pano_to_lidar(pano, lidar_K):
    """convert.py:140-156: [H,W] -> [N,3]."""
    return pano_to_lidar_with_intensities(pano, None, lidar_K)[:, :3]


def lidar_to_pano_with_intensities(local_points_with_intensities, lidar_H, lidar_W, lidar_K, max_depth=80):
    """convert.py:4-66: [N,4] points in the sensor frame -> (pano [H,W], intensities [H,W])."""
    pts = _f32(local_points_with_intensities, "local_points_with_intensities")
    if pts.dim() != 2 or pts.shape[1] != 4:
        raise ValueError("lidar_to_pano_with_intensities: expected [N, 4] points")
    H, W = int(lidar_H), int(lidar_W)
    dev = pts.device
    pano = torch.empty(H, W, dtype=torch.float32, device=dev)
    inten = torch.empty(H, W, dtype=torch.float32, device=dev)
    ws = torch.empty(max(1, _lib.lib().l4d_lidar_to_pano_workspace(H, W)), dtype=torch.uint8, device=dev)
    ops.call("l4d_lidar_to_pano", ops._p(pts), pts.shape[0], H, W, float(lidar_K[0]), float(lidar_K[1]), float(max_depth),
             ops._p(pano), ops._p(inten), ops._p(ws), ops._stream())
    return pano, inten


def lidar_to_pano(local_points, lidar_H, lidar_W, lidar_K, max_depth=80):
    """convert.py:69-96: [N,3] -> pano [H,W].  (The reference's version passes a misspelt keyword and cannot run.)"""
    pts = _f32(local_points, "local_points")
    pts4 = torch.cat([pts, torch.zeros_like(pts[:, :1])], dim=1)
    return lidar_to_pano_with_intensities(pts4, lidar_H, lidar_W, lidar_K, max_depth)[0]
