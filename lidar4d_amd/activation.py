"""``trunc_exp`` of the reference (model/activation.py:6-20): exp forward in fp32, backward
g * exp(clamp(x, -15, 15)).  Elementwise torch ops on the device; inside the fused render path the same pair is
folded into the sigma-network epilogue kernels (l4d_sigma_from_h / l4d_sigma_bwd)."""
import torch


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply
