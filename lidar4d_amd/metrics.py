"""Point-cloud evaluation on the device.  Mirror of the reference's ``fscore`` and ``PointsMeter``
(utils/metrics.py:13-27, 224-274): range image -> points (lidar4d_amd/convert.py instead of numpy) -> chamfer distance
(lidar4d_amd/chamfer.py instead of the CUDA extension) -> chamfer distance + F-score at 0.05 (squared-distance
threshold, as the reference uses it).  Nothing leaves the GPU until ``measure()``.
"""
import numpy as np
import torch

from .chamfer import chamfer_3DDist
from .convert import pano_to_lidar


def fscore(dist1, dist2, threshold=0.001):
    """utils/metrics.py:13-27.  dist1 [B, n] / dist2 [B, m]: SQUARED nearest-neighbour distances of the two directions
    (so ``threshold`` is a squared distance too).  Returns (F-score, precision, recall), each [B]; F is 0 where both
    precision and recall are 0."""
    precision = (dist1 < threshold).float().mean(dim=1)
    recall = (dist2 < threshold).float().mean(dim=1)
    denom = precision + recall
    f = torch.where(denom > 0, 2 * precision * recall / denom.clamp_min(1e-30), torch.zeros_like(denom))
    return f, precision, recall


class PointsMeter:
    def __init__(self, scale, intrinsics):
        self.V = []
        self.N = 0
        self.scale = scale
        self.intrinsics = intrinsics

    def clear(self):
        self.V = []
        self.N = 0

    def update(self, preds, truths):
        """preds, truths: [B, H, W] range images in scene units (depth * scale); only element 0 is used, like the
        reference (utils/metrics.py:253-254)."""
        preds = preds / self.scale
        truths = truths / self.scale
        pred_lidar = pano_to_lidar(preds[0], self.intrinsics)
        gt_lidar = pano_to_lidar(truths[0], self.intrinsics)
        dist1, dist2, _, _ = chamfer_3DDist()(pred_lidar[None], gt_lidar[None])
        if pred_lidar.shape[0] == 0 or gt_lidar.shape[0] == 0:
            # a degenerate frame (e.g. every predicted ray-drop <= 0.5 early in training): record the worst value of both
            # metrics instead of aborting the evaluation (the mean over an empty cloud would be nan)
            self.V.append(torch.tensor([float("inf"), 0.0], device=preds.device))
            self.N += 1
            return
        chamfer_dis = dist1.mean() + dist2.mean()
        f_score, _, _ = fscore(dist1, dist2, 0.05)
        self.V.append(torch.stack([chamfer_dis, f_score[0]]))  # stays on the device
        self.N += 1

    def measure(self):
        assert self.N == len(self.V)
        return torch.stack(self.V).mean(0).cpu().numpy().astype(np.float64)

    def report(self):
        return f"CD f-score = {self.measure()}"


class RaydropMeter:
    """utils/metrics.py:172-226: RMSE, accuracy and F1 of the ray-drop probability image against the 0/1 ground truth,
    thresholded at ``ratio``.  Torch on whatever device the inputs live on; one host transfer in ``measure()``."""

    def __init__(self, ratio=0.5):
        self.V = []
        self.N = 0
        self.ratio = ratio

    def clear(self):
        self.V = []
        self.N = 0

    def update(self, preds, truths):
        preds = torch.as_tensor(preds).detach().double()
        truths = torch.as_tensor(truths).detach().to(preds)
        rmse = ((truths - preds) ** 2).mean().sqrt()
        hit = (preds > self.ratio).to(preds)
        acc = (hit == truths).double().mean()
        tp = ((truths == 1) & (hit == 1)).sum().double()
        fp = ((truths == 0) & (hit == 1)).sum().double()
        fn = ((truths == 1) & (hit == 0)).sum().double()
        precision, recall = tp / (tp + fp), tp / (tp + fn)
        f1 = 2 * (precision * recall) / (precision + recall)  # nan when nothing is predicted / present, like the reference
        self.V.append(torch.stack([rmse, acc, f1]))
        self.N += 1

    def measure(self):
        assert self.N == len(self.V)
        return torch.stack(self.V).mean(0).cpu().numpy()

    def report(self):
        return f"Rdrop_error (RMSE, Acc, F1) = {self.measure()}"
