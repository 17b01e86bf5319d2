"""CPU restatement of the reference's range-image <-> point-cloud conversions (utils/convert.py).
TEST INFRASTRUCTURE ONLY -- pinned against tests/golden/convert.npz, which oracle/make_golden_next.py
produced by running the reference's own functions.

numpy semantics that matter for parity (NumPy >= 2 promotion: python scalars are weak, so float32 data
stays float32): directions and angles are computed in float32; ``int(round(.))`` rounds half to even.
"""
import numpy as np


def pano_to_lidar_with_intensities(pano, intensities, lidar_K):
    """utils/convert.py:99-137."""
    fov_up, fov = float(lidar_K[0]), float(lidar_K[1])  # python floats (weak): the float32 image decides the dtype
    H, W = pano.shape
    col, row = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    beta = -(col - W / 2) / W * 2 * np.pi
    alpha = (fov_up - row / H * fov) / 180 * np.pi
    dirs = np.stack([np.cos(alpha) * np.cos(beta), np.cos(alpha) * np.sin(beta), np.sin(alpha)], -1)
    pts = np.concatenate([dirs * pano.reshape(H, W, 1), intensities.reshape(H, W, 1)], axis=2)
    return pts[pano != 0.0]


def lidar_to_pano_with_intensities(points, lidar_H, lidar_W, lidar_K, max_depth=80):
    """utils/convert.py:4-66, vectorised: per pixel the closest point wins, the first of equally close ones stays."""
    points = np.asarray(points, dtype=np.float32)
    fov_up, fov = float(lidar_K[0]), float(lidar_K[1])
    fov_down = fov - fov_up
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    dist = np.sqrt((x * x + y * y) + z * z).astype(np.float32)
    beta = np.float32(np.pi) - np.arctan2(y, x)
    alpha = np.arctan2(z, np.sqrt(x * x + y * y)) + np.float32(fov_down / 180 * np.pi)
    c = np.rint(beta / np.float32(2 * np.pi / lidar_W)).astype(np.int64)
    r = np.rint(np.float32(lidar_H) - alpha / np.float32(fov / 180 * np.pi / lidar_H)).astype(np.int64)
    ok = (dist < max_depth) & (r >= 0) & (r < lidar_H) & (c >= 0) & (c < lidar_W) & (dist != 0)
    idx = np.nonzero(ok)[0]
    pix = r[idx] * lidar_W + c[idx]
    order = np.lexsort((idx, dist[idx], pix))          # by pixel, then range, then position in the array
    first = np.ones(len(order), dtype=bool)
    first[1:] = pix[order][1:] != pix[order][:-1]
    win = idx[order][first]
    pano = np.zeros(lidar_H * lidar_W, dtype=np.float32)
    inten = np.zeros(lidar_H * lidar_W, dtype=np.float32)
    pano[pix[order][first]] = dist[win]
    inten[pix[order][first]] = points[win, 3]
    return pano.reshape(lidar_H, lidar_W), inten.reshape(lidar_H, lidar_W)
