"""``trunc_exp``: the density activation of LiDAR4D (reference model/activation.py:6-20) -- exp evaluated in fp32 whose
backward uses exp(clamp(x, -15, 15)), so a huge pre-activation cannot blow the gradient up.

This module serves callers of the operator-level API (``LiDAR4D.density``).  Inside the fused render path the same pair
lives in the sigma-network epilogue kernels (``l4d_sigma_from_h`` / ``l4d_sigma_bwd``, lidar4d_amd/csrc/render.hip).
"""
import math

import torch

_LIMIT = 15.0


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre):
        pre32 = pre.to(torch.float32)
        out = pre32.exp()
        ctx.save_for_backward(pre32, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        pre32, out = ctx.saved_tensors
        # exp(clamp(x, -L, L)): reuse the forward value inside the window, the window edge outside of it
        edge = torch.where(pre32 > 0, math.exp(_LIMIT), math.exp(-_LIMIT))
        slope = torch.where(pre32.abs() <= _LIMIT, out, edge.to(out.dtype))
        return grad_out * slope


def trunc_exp(x):
    return _TruncExp.apply(x)
