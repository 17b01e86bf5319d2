"""Oracle restatement of the tiny-cuda-nn operators used by LiDAR4D.  TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  tiny-cuda-nn (NVlabs, PyTorch extension ``tinycudann``) is an un-vendored,
unpinned dependency of the reference (reference README.md:88-91 clones ``master``); it is not in
/root/reference and cannot be installed here.  This file restates its *published* algorithm as
recorded in SURVEY.md Appendix A (``encodings/grid.h``, ``encodings/frequency.h``,
``src/fully_fused_mlp.cu``, ``bindings/torch/tinycudann/modules.py``) and presents it behind the
same Python surface the reference binds (``Encoding``, ``Network``; reference call sites
model/hash_field.py:47-57,107-117, model/flow_field.py:67-77, model/lidar4d.py:68-117).

Precision modes (``set_precision``):

``"tcnn"``  the rounding points of tiny-cuda-nn's fp16 build, which the HIP path reproduces:
            (R1) parameters fp32 -> fp16 on every forward; (R2) every Encoding output -> fp16;
            (R3) Network input -> fp16 (after padding with 1.0), every hidden activation -> fp16
            after ReLU, output -> fp16.  All *accumulation* is fp32 (tiny-cuda-nn itself
            accumulates in fp16; fp32 is strictly closer to the real-valued result).
            Output tensors are fp16, as tiny-cuda-nn's are.
``"fp32"``  no rounding anywhere, fp32 outputs: the idealised reference used to generate the
            golden fixtures through the reference's own glue code and to bound the fp16 effect.

Deliberate deviation (documented): the Frequency encoding is evaluated as the exact
``sin(2^k*pi*x)`` / ``cos(2^k*pi*x)`` (fp64 here, exact range reduction + sinpi/cospi in the HIP
kernel); tiny-cuda-nn uses the fast-math ``__sinf`` whose error at the top octaves is ~1e-3.
"""
import math

import numpy as np
import torch
import torch.nn as nn

_PRECISION = "tcnn"

PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)


def set_precision(mode):
    global _PRECISION
    assert mode in ("tcnn", "fp32")
    _PRECISION = mode


def get_precision():
    return _PRECISION


# Gradient precision.  "fp32" (default): adjoints and parameter-gradient accumulation in fp32 -- the exact gradient of the
# (fp16-rounded) forward, what the HIP path is compared with elementwise.  "tcnn16": what tiny-cuda-nn's fp16 build does on the
# way back (SURVEY A.1 "Backward wrt params", A.3): every adjoint that crosses a rounding point is an fp16 number in the
# loss-scaled domain, and a hash table's gradient is ACCUMULATED IN fp16 -- one atomicAdd(__half2) per corner, each add rounded
# to fp16 -- then un-scaled.  Not a parity target: it measures how far tiny-cuda-nn's own arithmetic sits from the exact gradient,
# the yardstick for the HIP path's fp16 adjoints (tests/test_gpu_c3_parity.py::test_gradient_error_vs_tcnn_fp16_accumulation).
_GRAD_MODE = "fp32"
_GRAD_SCALE = 128.0  # the scale fp16 adjoints carry: tiny-cuda-nn's loss_scale (128) x the GradScaler's scale


_ADJ_MAX = [0.0]     # grad mode "probe": largest |adjoint| that crossed a rounding point (un-scaled), same arithmetic as "fp32"


def set_grad_precision(mode, scale=128.0):
    global _GRAD_MODE, _GRAD_SCALE
    assert mode in ("fp32", "tcnn16", "probe")
    _GRAD_MODE, _GRAD_SCALE = mode, float(scale)
    if mode == "probe":
        _ADJ_MAX[0] = 0.0


def get_grad_precision():
    return _GRAD_MODE, _GRAD_SCALE


def probed_adjoint_max():
    """Largest |adjoint| seen at a rounding point since set_grad_precision("probe"): 65504 / this bounds the loss scale under
    which tiny-cuda-nn's fp16 backward stays finite (what a GradScaler would back off to)."""
    return _ADJ_MAX[0]


def _round_adjoint(g):
    if _GRAD_MODE == "tcnn16":
        return (g * _GRAD_SCALE).half().float() / _GRAD_SCALE
    if _GRAD_MODE == "probe" and g.numel():
        _ADJ_MAX[0] = max(_ADJ_MAX[0], float(g.abs().max()))
    return g


class _RoundHalfSTE(torch.autograd.Function):
    """fp32 -> fp16 -> fp32 rounding, identity in backward (straight-through; in grad mode "tcnn16" the adjoint is an fp16
    number in the loss-scaled domain, like the tensors tiny-cuda-nn's backward hands from layer to layer)."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return _round_adjoint(g)


def fp16_scatter_accumulate(n_entries, idx, contrib):
    """sum of contrib [N, F] into rows idx [N] of an [n_entries, F] table where EVERY add is rounded to fp16 (sequentially, in
    the given order per entry): tiny-cuda-nn's atomicAdd(__half2) accumulation of a hash table's gradient.  Returns fp32."""
    c16 = contrib.half()
    order = torch.argsort(idx, stable=True)
    idx_s, c_s = idx[order], c16[order]
    n = idx_s.numel()
    first = torch.ones(n, dtype=torch.bool)
    first[1:] = idx_s[1:] != idx_s[:-1]
    start = torch.cummax(torch.where(first, torch.arange(n), torch.zeros(n, dtype=torch.int64)), 0).values
    rank = torch.arange(n) - start
    acc = torch.zeros(n_entries, contrib.shape[1], dtype=torch.float16)
    for r in range(int(rank.max()) + 1 if n else 0):
        sel = rank == r
        e = idx_s[sel]
        acc[e] = (acc[e].float() + c_s[sel].float()).half()  # the exact sum of two halfs, rounded to half: what a half add gives
    return acc.float()


class _LevelInterpFp16Grad(torch.autograd.Function):
    """One hash-grid level, out = sum_c w_c * table[idx_c], whose backward is tiny-cuda-nn's: dL/dy arrives as fp16 (loss-scaled),
    each corner contributes half((float)dy * w), contributions are accumulated into an fp16 gradient table (grid.h backward)."""

    @staticmethod
    def forward(ctx, table, idx, w):
        ctx.save_for_backward(idx, w)
        ctx.n = table.shape[0]
        return (w.unsqueeze(-1) * table[idx]).sum(1)

    @staticmethod
    def backward(ctx, d_out):
        idx, w = ctx.saved_tensors
        dy16 = (d_out * _GRAD_SCALE).half().float()                      # [P, F]
        contrib = (dy16.unsqueeze(1) * w.unsqueeze(-1))                   # [P, C, F] fp32 products, rounded to half below
        g = fp16_scatter_accumulate(ctx.n, idx.reshape(-1), contrib.reshape(-1, contrib.shape[-1]))
        return g / _GRAD_SCALE, None, None


# Forward jitter (experiment / yardstick): after an ACTIVATION is rounded to fp16, a fraction of the elements is moved by one fp16
# ulp up or down -- what a different but equally legitimate fp32 summation order does to the last bit (the HIP kernels and this
# oracle agree bit for bit on >= 98 % of the sigma logits and differ by one ulp on the rest).  The change of the
# exact gradient under such a jitter is the floor below which no elementwise gradient comparison between two implementations
# of the same fp16 network can go.
_JITTER = {"frac": 0.0, "gen": None}


def set_forward_jitter(frac, seed=0):
    _JITTER["frac"] = float(frac)
    _JITTER["gen"] = torch.Generator().manual_seed(seed) if frac > 0 else None


class _JitterSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        h = x.half()
        r = torch.rand(h.shape, generator=_JITTER["gen"])
        up = r < 0.5 * _JITTER["frac"]
        dn = (r >= 0.5 * _JITTER["frac"]) & (r < _JITTER["frac"])
        bits = h.view(torch.int16).to(torch.int32)
        pos = bits >= 0  # sign bit clear: incrementing the bit pattern moves away from zero
        step = torch.where(up, torch.where(pos, 1, -1), torch.where(dn, torch.where(pos, -1, 1), 0))
        ok = (bits & 0x7FFF) > 0x0400  # leave zeros / subnormals alone (ReLU zeros must stay zeros)
        ok &= (bits & 0x7FFF) < 0x7BFF
        bits = torch.where(ok, bits + step, bits)
        return bits.to(torch.int16).view(torch.float16).float()

    @staticmethod
    def backward(ctx, g):
        return _round_adjoint(g)


def rh(x):
    """Apply an fp16 rounding point when the oracle is in tcnn mode."""
    if _PRECISION == "tcnn":
        if _JITTER["frac"] > 0.0 and not isinstance(x, nn.Parameter) and x.requires_grad and x.dim() == 2:
            return _JitterSTE.apply(x)
        return _RoundHalfSTE.apply(x)
    return x


def _out(x):
    """tcnn modules emit fp16 tensors; keep autograd alive through the cast."""
    if _PRECISION == "tcnn":
        return x.half()
    return x


# ----------------------------------------------------------------------------------------------
# HashGrid
# ----------------------------------------------------------------------------------------------
def grid_level_meta(n_dims, n_levels, log2_hashmap_size, base_resolution, per_level_scale):
    """Per-level (scale, resolution, n_entries, entry offset, hashed?) exactly as tiny-cuda-nn's host
    code derives them, in float32 arithmetic."""
    f32 = np.float32
    log2_pls = np.log2(f32(per_level_scale)).astype(f32)
    scales, ress, sizes, offsets, hashed = [], [], [], [], []
    off = 0
    for lvl in range(n_levels):
        scale = f32(np.exp2(f32(lvl) * log2_pls).astype(f32) * f32(base_resolution) - f32(1.0))
        res = int(np.ceil(scale)) + 1
        max_params = (2 ** 32 - 1) // 2
        dense = res ** n_dims
        n = max_params if float(dense) > float(max_params) else dense
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        # tiny-cuda-nn grid_index(): stride walk decides dense vs hashed addressing
        stride = 1
        for _ in range(n_dims):
            if stride > n:
                break
            stride = (stride * res) & 0xFFFFFFFF
        scales.append(float(scale))
        ress.append(res)
        sizes.append(n)
        offsets.append(off)
        hashed.append(bool(n < stride))
        off += n
    return {
        "scale": scales,
        "res": ress,
        "size": sizes,
        "offset": offsets,
        "hashed": hashed,
        "n_entries": off,
    }


def hashgrid_corner_indices(x, meta, lvl, n_dims):
    """Corner entry indices [P, 2^D] (int64, level-local) and weights [P, 2^D] (fp32) of one level."""
    scale = torch.tensor(meta["scale"][lvl], dtype=torch.float32)
    res = meta["res"][lvl]
    n = meta["size"][lvl]
    # fmaf(scale, x, 0.5): the fp32 x fp32 product is exact in fp64, so this is the fused result
    pos = (x.double() * scale.double() + 0.5).float()
    cell = torch.floor(pos)
    frac = pos - cell
    cell_u = cell.to(torch.int64) & 0xFFFFFFFF  # (uint32)(int)floor: negatives wrap
    idx_list, w_list = [], []
    for c in range(1 << n_dims):
        w = torch.ones_like(frac[:, 0])
        g = []
        for d in range(n_dims):
            if (c >> d) & 1:
                w = w * frac[:, d]
                g.append((cell_u[:, d] + 1) & 0xFFFFFFFF)
            else:
                w = w * (1.0 - frac[:, d])
                g.append(cell_u[:, d])
        if meta["hashed"][lvl]:
            h = torch.zeros_like(g[0])
            for d in range(n_dims):
                h = h ^ ((g[d] * PRIMES[d]) & 0xFFFFFFFF)
            idx = h
        else:
            idx = torch.zeros_like(g[0])
            stride = 1
            for d in range(n_dims):
                if stride > n:
                    break
                idx = (idx + g[d] * stride) & 0xFFFFFFFF
                stride = (stride * res) & 0xFFFFFFFF
        idx_list.append(idx % n)
        w_list.append(w)
    return torch.stack(idx_list, -1), torch.stack(w_list, -1)


class HashGridRef(nn.Module):
    """tcnn ``Encoding`` with ``otype: HashGrid`` (Hash grid type, Linear interpolation)."""

    def __init__(self, n_input_dims, cfg, seed=1337):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_levels = int(cfg["n_levels"])
        self.n_features = int(cfg["n_features_per_level"])
        self.meta = grid_level_meta(
            n_input_dims,
            self.n_levels,
            int(cfg["log2_hashmap_size"]),
            int(cfg["base_resolution"]),
            float(cfg["per_level_scale"]),
        )
        self.n_output_dims = self.n_levels * self.n_features
        g = torch.Generator().manual_seed(seed)
        p = (torch.rand(self.meta["n_entries"] * self.n_features, generator=g) * 2 - 1) * 1e-4
        self.params = nn.Parameter(p)
        self.loss_scale = 128.0
        self.seed = seed

    def forward(self, x):
        x = x.to(torch.float32).contiguous()
        table = rh(self.params).view(-1, self.n_features)
        outs = []
        for lvl in range(self.n_levels):
            idx, w = hashgrid_corner_indices(x, self.meta, lvl, self.n_input_dims)
            if _GRAD_MODE == "tcnn16" and torch.is_grad_enabled() and self.params.requires_grad:
                o = self.meta["offset"][lvl]
                outs.append(_LevelInterpFp16Grad.apply(table[o:o + self.meta["size"][lvl]], idx, w))
                continue
            vals = table[self.meta["offset"][lvl] + idx]  # [P, 2^D, F]
            outs.append((w.unsqueeze(-1) * vals).sum(1))
        return _out(rh(torch.cat(outs, -1)))


# ----------------------------------------------------------------------------------------------
# Frequency
# ----------------------------------------------------------------------------------------------
class FrequencyRef(nn.Module):
    """tcnn ``Encoding`` with ``otype: Frequency``: per input dim
    [sin(2^0 pi x), cos(2^0 pi x), ..., sin(2^11 pi x), cos(2^11 pi x)]; ``degree`` is ignored."""

    def __init__(self, n_input_dims, cfg):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.n_frequencies = int(cfg.get("n_frequencies", 12))
        self.n_output_dims = n_input_dims * self.n_frequencies * 2
        self.params = nn.Parameter(torch.zeros(0))
        self.loss_scale = 128.0
        self.seed = 1337

    def forward(self, x):
        x = x.to(torch.float32)
        xd = x.double()
        k = torch.arange(self.n_frequencies, dtype=torch.float64)
        arg = xd.unsqueeze(-1) * torch.exp2(k) * math.pi  # [P, D, K]
        out = torch.stack([torch.sin(arg), torch.cos(arg)], -1)  # [P, D, K, 2]
        out = out.reshape(x.shape[0], -1).float()
        return _out(rh(out))


# ----------------------------------------------------------------------------------------------
# FullyFusedMLP
# ----------------------------------------------------------------------------------------------
def mlp_layer_shapes(n_in, n_out, n_neurons, n_hidden_layers):
    """[(rows, cols)] of the row-major weight matrices in tcnn's flat ``params`` order."""
    in_pad = (n_in + 15) // 16 * 16
    out_pad = (n_out + 15) // 16 * 16
    shapes = [(n_neurons, in_pad)]
    for _ in range(n_hidden_layers - 1):
        shapes.append((n_neurons, n_neurons))
    shapes.append((out_pad, n_neurons))
    return shapes


class FusedMLPRef(nn.Module):
    """tcnn ``Network`` with ``otype: FullyFusedMLP``, ReLU hidden activation, no output activation,
    no biases; input padded to a multiple of 16 with the constant 1.0."""

    def __init__(self, n_input_dims, n_output_dims, cfg, seed=1337):
        super().__init__()
        assert cfg.get("activation", "ReLU") == "ReLU"
        assert cfg.get("output_activation", "None") == "None"
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.n_neurons = int(cfg["n_neurons"])
        self.n_hidden_layers = int(cfg["n_hidden_layers"])
        self.shapes = mlp_layer_shapes(n_input_dims, n_output_dims, self.n_neurons, self.n_hidden_layers)
        g = torch.Generator().manual_seed(seed)
        chunks = []
        for rows, cols in self.shapes:
            bound = math.sqrt(6.0 / (rows + cols))
            chunks.append((torch.rand(rows * cols, generator=g) * 2 - 1) * bound)
        self.params = nn.Parameter(torch.cat(chunks))
        self.loss_scale = 128.0
        self.seed = seed

    def weights(self):
        ws, off = [], 0
        p = rh(self.params)
        for rows, cols in self.shapes:
            ws.append(p[off:off + rows * cols].view(rows, cols))
            off += rows * cols
        return ws

    def forward(self, x):
        x = x.to(torch.float32)
        in_pad = self.shapes[0][1]
        if in_pad > x.shape[1]:
            x = torch.cat([x, torch.ones(x.shape[0], in_pad - x.shape[1], dtype=x.dtype)], -1)
        h = rh(x)
        ws = self.weights()
        for w in ws[:-1]:
            h = rh(torch.relu(h @ w.t()))
        y = rh(h @ ws[-1].t())
        return _out(y[:, : self.n_output_dims])


# ----------------------------------------------------------------------------------------------
# tinycudann Python surface
# ----------------------------------------------------------------------------------------------
def Encoding(n_input_dims, encoding_config, dtype=None, seed=1337):
    otype = encoding_config["otype"]
    if otype == "HashGrid":
        return HashGridRef(n_input_dims, encoding_config, seed=seed)
    if otype == "Frequency":
        return FrequencyRef(n_input_dims, encoding_config)
    raise ValueError(f"oracle: unsupported encoding otype {otype!r}")


def Network(n_input_dims, n_output_dims, network_config, seed=1337):
    if network_config["otype"] != "FullyFusedMLP":
        raise ValueError("oracle: only FullyFusedMLP is restated")
    return FusedMLPRef(n_input_dims, n_output_dims, network_config, seed=seed)
