"""Oracle for the chamfer distance (reference utils/chamfer3D/chamfer3D.cu:11-194): brute-force torch.  TEST
INFRASTRUCTURE ONLY.  The reference kernel cannot run in the build container (no GPU) and has no fixtures, so this
restatement follows its arithmetic -- squared distance dx*dx + dy*dy + dz*dz in fp32, first minimum wins, backward
2 * g * (p - q) into both clouds -- and is "parity unpinned" against the CUDA binary (nvcc contracts the sum into
FMAs, which can move a distance by one ulp and flip near-ties)."""
import torch


def chamfer(xyz1, xyz2):
    """xyz1 [B,N,3], xyz2 [B,M,3] -> dist1 [B,N], dist2 [B,M], idx1, idx2 (int64)."""
    d = xyz1.unsqueeze(2) - xyz2.unsqueeze(1)           # [B,N,M,3]
    dd = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]
    dist1, idx1 = dd.min(2)
    dist2, idx2 = dd.min(1)
    return dist1, dist2, idx1, idx2
