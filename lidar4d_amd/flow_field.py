"""Scene-flow field.  Mirror of the reference's model/flow_field.py:40-130 (``FlowField`` with use_grid=True,
use_freq=False): 3-D hash grid (8 levels x 8 features) -> interpT over feature chunks -> bias-free ReLU MLP
16 -> 64 -> 64 -> 6.  State-dict keys ``grid_enc.params``, ``mlp.{0,2,4}.weight`` as in the reference.

Two launches: grid lookup + interpT fused (l4d_hashgrid_t_fwd with one table), then the MFMA MLP (l4d_mlp_fwd)
with the fp16-operand / fp32-accumulate numerics the reference's autocast run gives its nn.Linear layers.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from . import tcnn


class _FlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xt, mod, grid_params, *weights):
        xt_c = xt.detach().to(torch.float32).contiguous()
        t_dev = xt_c[0, 3:4]
        xf = ops.hashgrid_t_fwd(mod.grid_enc.meta, xt_c, (0, 1, 2), [mod.grid_enc._half_params()], t_dev, half_out=True)
        w16 = mod._weights16()
        y, act = ops.mlp_fwd(xf, w16, mod.n_hidden, save_act=True)
        ctx.mod = mod
        ctx.save_for_backward(xt_c, xf, act, w16)
        # fp32 tensor holding the fp16-rounded outputs (same values as the reference's autocast nn.Linear): autograd hands a
        # tensor's gradient over in that tensor's dtype, and an fp16 dy is already saturated to inf at loss scale x |dy| > 65504
        # before backward() below can normalise it
        return y[:, :6].float()

    @staticmethod
    def backward(ctx, dy):
        xt_c, xf, act, w16 = ctx.saved_tensors
        mod = ctx.mod
        P = xt_c.shape[0]
        # fp16 adjoints under a power-of-two scale chosen ON THE DEVICE from this call's own upstream gradient (largest |dy| lands
        # in [2^11, 2^12): headroom for the two 64-term contractions behind it): the chamfer sums of the scene-flow loss hand over gradients of order 10^2 per point, which under a
        # constant 128 on top of the caller's GradScaler scale left the fp16 range at scale 4 and dragged the whole step's loss
        # scale down with it (tools/scale_probe.py).  The adjoint is linear in dy, so the kernels run unscaled (inv = 1) into
        # private buffers and the power of two is taken out again in fp32.  A non-finite dy gives a non-finite scale, hence
        # non-finite gradients: the overflow still reaches the scaler, it is never clipped (csrc/common.h f2h_grad).
        dyf = dy.float()
        amax = ops.absmax(dyf.contiguous()).reshape(())  # (one HIP launch, not a torch reduction: ops.absmax)
        k = torch.floor(torch.log2(4096.0 / amax.clamp_min(1e-30))).clamp(-40.0, 40.0)
        s, inv = torch.exp2(k), torch.exp2(-k)
        dy16 = torch.zeros(P, 16, dtype=torch.float16, device=dy.device)
        dy16[:, :6] = dyf * s
        gw = torch.zeros(w16.numel(), dtype=torch.float32, device=dy.device)
        dxf = ops.mlp_bwd(xf, act, dy16, w16, mod.n_hidden, gw, 1.0)
        ggrid = torch.zeros_like(mod.grid_enc.params)
        ops.hashgrid_t_bwd(mod.grid_enc.meta, xt_c, (0, 1, 2), 1, xt_c[0, 3:4], dxf, [ggrid], 1.0)
        ggrid *= inv
        gw *= inv
        return (None, None, ggrid) + tuple(mod._split_weight_grads(gw))


class FlowField(nn.Module):
    def __init__(self, input_dim=4, num_layers=3, hidden_dim=64, use_freq=False, num_freqs=6, use_grid=True,
                 num_basis=4, n_levels=8, n_features_per_level=8, base_resolution=32, max_resolution=8192,
                 log2_hashmap_size=18):
        super().__init__()
        if use_freq or not use_grid or num_basis != 4 or n_features_per_level != 8 or hidden_dim != 64:
            raise ValueError("FlowField: only the reference configuration (grid encoding, 8 features/level, "
                             "num_basis 4, hidden 64) is implemented")
        if not 2 <= num_layers <= 4:
            raise ValueError("FlowField: 2..4 layers supported")
        self.use_freq, self.use_grid = use_freq, use_grid
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        self.grid_enc = tcnn.Encoding(n_input_dims=3, encoding_config={
            "otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
            "per_level_scale": per_level_scale})
        self.n_levels, self.n_features_per_level, self.num_basis = n_levels, n_features_per_level, num_basis
        self.input_dim = self.grid_enc.n_output_dims // num_basis
        if self.input_dim % 16:
            raise ValueError("FlowField: grid width / num_basis must be a multiple of 16")
        layers = []
        for l in range(num_layers):
            i = self.input_dim if l == 0 else hidden_dim
            o = 6 if l == num_layers - 1 else hidden_dim
            layers.append(nn.Linear(i, o, bias=False))
            if l != num_layers - 1:
                layers.append(nn.ReLU())
        self.mlp = nn.Sequential(*layers)
        torch.nn.init.normal_(self.mlp[-1].weight.data, 0, 0.001)
        self.n_hidden = num_layers - 1
        self.loss_scale = 128.0

    def linears(self):
        return [m for m in self.mlp if isinstance(m, nn.Linear)]

    def weight_numel16(self):
        return 64 * self.input_dim + (self.n_hidden - 1) * 64 * 64 + 16 * 64

    def _weights16(self, out=None):
        """fp16 weights in the MLP kernel's layout: [64,in], (n_hidden-1) x [64,64], [16,64] (6 real rows)."""
        lins = self.linears()
        if out is None:
            out = torch.zeros(self.weight_numel16(), dtype=torch.float16, device=lins[0].weight.device)
        off = 0
        for m in lins[:-1]:
            n = m.weight.numel()
            out[off:off + n] = m.weight.detach().reshape(-1)
            off += n
        out[off:off + 6 * 64] = lins[-1].weight.detach().reshape(-1)
        return out

    def _split_weight_grads(self, gw):
        lins = self.linears()
        grads, off = [], 0
        for m in lins[:-1]:
            n = m.weight.numel()
            grads.append(gw[off:off + n].view_as(m.weight))
            off += n
        grads.append(gw[off:off + 6 * 64].view_as(lins[-1].weight))
        return grads

    def forward(self, xt):
        return _FlowFn.apply(xt, self, self.grid_enc.params, *[m.weight for m in self.linears()])
