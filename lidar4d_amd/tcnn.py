"""HIP-backed stand-ins for the two tiny-cuda-nn entry points LiDAR4D binds: ``Encoding`` and ``Network``.

Same surface as ``tinycudann`` (reference call sites model/hash_field.py:47-57,107-117,
model/flow_field.py:67-77, model/lidar4d.py:68-117): ``nn.Module``s with a flat fp32 ``.params``,
``.n_input_dims``, ``.n_output_dims``, ``.loss_scale``; ``forward(x[N, in]) -> fp16 [N, n_output_dims]``.
Kernels: lidar4d_amd/csrc/hashgrid.hip, render.hip (frequency), mlp.hip.  No CPU path.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .gridmeta import GridMeta


class _HalfParams:
    """fp16 compute copy of a module's fp32 ``params`` -- from the model's ParamStore when the module is part
    of a flattened LiDAR4D, else a private copy refreshed when the parameter's version changes."""

    def _half_params(self):
        store = getattr(self, "_store", None)
        if store is not None:
            return store.half(self.params)
        key = (self.params.data_ptr(), self.params._version)
        if getattr(self, "_p16_key", None) != key:
            self._p16 = ops.cast_f32_to_f16(self.params.detach().contiguous())
            self._p16_key = key
        return self._p16


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, mod):
        x = x.detach().to(torch.float32).contiguous()
        out = ops.hashgrid_fwd(mod.meta, x, range(mod.n_input_dims), mod._half_params())
        ctx.mod = mod
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        grad = torch.zeros_like(mod.params)
        ops.hashgrid_bwd(mod.meta, x, range(mod.n_input_dims), dout.contiguous(), grad, 1.0)
        return None, grad, None  # d/dx is not needed anywhere on the LiDAR4D path


class HashGridEncoding(nn.Module, _HalfParams):
    def __init__(self, n_input_dims, cfg, seed=1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.meta = GridMeta(n_input_dims, cfg["n_levels"], cfg["n_features_per_level"], cfg["log2_hashmap_size"],
                             cfg["base_resolution"], cfg["per_level_scale"])
        self.n_output_dims = self.meta.n_output_dims
        g = torch.Generator().manual_seed(seed)
        self.params = nn.Parameter((torch.rand(self.meta.n_params, generator=g) * 2 - 1) * 1e-4)
        self.loss_scale = 128.0
        self.seed = seed

    def forward(self, x):
        return _HashGridFn.apply(x, self.params, self)


class FrequencyEncoding(nn.Module):
    """``otype: Frequency`` -- 12 octaves of sin/cos per input dimension, no parameters (``degree`` is ignored
    by tiny-cuda-nn)."""

    def __init__(self, n_input_dims, cfg):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_frequencies = int(cfg.get("n_frequencies", 12))
        self.n_output_dims = n_input_dims * self.n_frequencies * 2
        self.params = nn.Parameter(torch.zeros(0))
        self.loss_scale = 128.0
        self.seed = 1337

    def forward(self, x):
        return ops.freq_fwd(x.detach().to(torch.float32).contiguous(), self.n_frequencies)


def Encoding(n_input_dims, encoding_config, dtype=None, seed=1337):
    otype = encoding_config["otype"]
    if otype == "HashGrid":
        return HashGridEncoding(n_input_dims, encoding_config, seed=seed)
    if otype == "Frequency":
        return FrequencyEncoding(n_input_dims, encoding_config)
    raise ValueError(f"lidar4d_amd.tcnn: unsupported encoding otype {otype!r} (LiDAR4D uses HashGrid and Frequency)")


def mlp_layer_shapes(n_in, n_out, n_neurons, n_hidden_layers):
    in_pad = (n_in + 15) // 16 * 16
    out_pad = (n_out + 15) // 16 * 16
    return [(n_neurons, in_pad)] + [(n_neurons, n_neurons)] * (n_hidden_layers - 1) + [(out_pad, n_neurons)]


class _MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, mod):
        P = x.shape[0]
        x16 = torch.ones(P, mod.in_pad, dtype=torch.float16, device=x.device)
        x16[:, : mod.n_input_dims] = x.detach()
        y, act = ops.mlp_fwd(x16, mod._half_params(), mod.n_hidden_layers, save_act=True)
        ctx.mod = mod
        ctx.save_for_backward(x16, act)
        ctx.x_dtype = x.dtype
        return y[:, : mod.n_output_dims]

    @staticmethod
    def backward(ctx, dy):
        x16, act = ctx.saved_tensors
        mod = ctx.mod
        P = x16.shape[0]
        s = mod.loss_scale
        dy16 = torch.zeros(P, 16, dtype=torch.float16, device=dy.device)
        dy16[:, : mod.n_output_dims] = dy.float() * s  # not saturated: an overflow becomes inf and reaches the scaler (csrc/common.h f2h_grad)
        grad = torch.zeros_like(mod.params)
        dx16 = ops.mlp_bwd(x16, act, dy16, mod._half_params(), mod.n_hidden_layers, grad, 1.0 / s)
        dx = (dx16[:, : mod.n_input_dims].float() / s).to(ctx.x_dtype)
        return dx, grad, None


class FullyFusedMLP(nn.Module, _HalfParams):
    """``otype: FullyFusedMLP``: bias-free, ReLU hidden layers of width 64, no output activation; the input is
    padded to a multiple of 16 with the constant 1.0, the output to 16."""

    def __init__(self, n_input_dims, n_output_dims, cfg, seed=1337):
        super().__init__()
        if cfg.get("activation", "ReLU") != "ReLU" or cfg.get("output_activation", "None") != "None":
            raise ValueError("lidar4d_amd.tcnn.Network: only ReLU / no output activation (what LiDAR4D configures)")
        if int(cfg["n_neurons"]) != 64 or n_output_dims > 16:
            raise ValueError("lidar4d_amd.tcnn.Network: n_neurons must be 64 and n_output_dims <= 16")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.n_hidden_layers = int(cfg["n_hidden_layers"])
        self.shapes = mlp_layer_shapes(n_input_dims, n_output_dims, 64, self.n_hidden_layers)
        self.in_pad = self.shapes[0][1]
        if self.in_pad not in (16, 32, 64, 96, 128, 160, 176, 192) or not 1 <= self.n_hidden_layers <= 3:
            raise ValueError("lidar4d_amd.tcnn.Network: padded input width in {16,32,64,96,128,160,176,192} and 1..3 "
                             "hidden layers are supported")
        g = torch.Generator().manual_seed(seed)
        chunks = []
        for rows, cols in self.shapes:
            bound = math.sqrt(6.0 / (rows + cols))
            chunks.append((torch.rand(rows * cols, generator=g) * 2 - 1) * bound)
        self.params = nn.Parameter(torch.cat(chunks))
        self.loss_scale = 128.0
        self.seed = seed

    def forward(self, x):
        return _MLPFn.apply(x, self.params, self)


def Network(n_input_dims, n_output_dims, network_config, seed=1337):
    if network_config["otype"] != "FullyFusedMLP":
        raise ValueError("lidar4d_amd.tcnn.Network: only FullyFusedMLP is implemented (what LiDAR4D configures)")
    return FullyFusedMLP(n_input_dims, n_output_dims, network_config, seed=seed)
