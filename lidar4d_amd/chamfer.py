"""Chamfer distance.  Mirror of the reference's utils/chamfer3D/dist_chamfer_3D.py:31-83 (``chamfer_3DDist`` module and
its autograd function) on the HIP kernels of lidar4d_amd/csrc/chamfer.hip (which replace utils/chamfer3D/chamfer3D.cu)."""
import torch
from torch import nn
from torch.autograd import Function

from . import _lib, ops


class chamfer_3DFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        if xyz1.dim() != 3 or xyz2.dim() != 3 or xyz1.shape[-1] != 3 or xyz2.shape[-1] != 3 or xyz1.shape[0] != xyz2.shape[0]:
            raise ValueError(f"chamfer_3DDist expects clouds [B, n, 3] and [B, m, 3]; got {tuple(xyz1.shape)} and {tuple(xyz2.shape)}")
        batchsize, n, m = xyz1.shape[0], xyz1.shape[1], xyz2.shape[1]
        xyz1 = xyz1.detach().float().contiguous()
        xyz2 = xyz2.detach().float().contiguous()
        ops._chk(xyz1, torch.float32, "xyz1"), ops._chk(xyz2, torch.float32, "xyz2")
        dev = xyz1.device
        dist1 = torch.empty(batchsize, n, device=dev)
        dist2 = torch.empty(batchsize, m, device=dev)
        idx1 = torch.empty(batchsize, n, dtype=torch.int32, device=dev)
        idx2 = torch.empty(batchsize, m, dtype=torch.int32, device=dev)
        ctx.degenerate = n == 0 or m == 0
        if ctx.degenerate:
            # One cloud is empty (early in training every predicted ray-drop can be <= 0.5 -> no predicted points): the
            # reference kernel has nothing to index; report "infinitely far" for the non-empty side so that the caller's
            # metrics degrade (chamfer inf, F-score 0) instead of the evaluation aborting.  No gradient flows.
            dist1.fill_(float("inf")), dist2.fill_(float("inf")), idx1.zero_(), idx2.zero_()
            ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
            ctx.mark_non_differentiable(idx1, idx2)
            return dist1, dist2, idx1, idx2
        ws = torch.empty(_lib.lib().l4d_chamfer_workspace(batchsize, n, m), dtype=torch.uint8, device=dev)
        ops.call("l4d_chamfer_fwd", ops._p(xyz1), ops._p(xyz2), batchsize, n, m, ops._p(dist1), ops._p(dist2), ops._p(idx1),
                 ops._p(idx2), ops._p(ws), ops._stream())
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        graddist1 = graddist1.float().contiguous()
        graddist2 = graddist2.float().contiguous()
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2)
        if ctx.degenerate:
            return gradxyz1, gradxyz2
        ops.call("l4d_chamfer_bwd", ops._p(xyz1), ops._p(xyz2), b, n, m, ops._p(graddist1), ops._p(graddist2), ops._p(idx1),
                 ops._p(idx2), ops._p(gradxyz1), ops._p(gradxyz2), ops._stream())
        return gradxyz1, gradxyz2


class chamfer_3DDist(nn.Module):
    def forward(self, input1, input2):
        return chamfer_3DFunction.apply(input1.contiguous(), input2.contiguous())
