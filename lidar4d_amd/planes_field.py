"""Hex-plane (K-Planes) field.  Mirror of the reference's model/planes_field.py:144-239 (``Planes4D``): same
constructor arguments, ``planes`` ModuleList-of-ParameterList with [1, C, R_b, R_a] parameters (time planes
[1, C, 8, R]), same init (static U(0.1, 0.5), time planes ones; planes_field.py:48-51).

One kernel samples all planes of all scales (bilinear, align_corners=True, border), takes the per-scale
products and concatenates scales (l4d_planes_fwd / l4d_planes_bwd), instead of 24 F.grid_sample launches.
"""
import itertools

import torch
import torch.nn as nn

from . import ops
from .params import _EPOCH


class _PlanesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xt, mod, which, *planes):
        xt_c = xt.detach().to(torch.float32).contiguous()
        arena = mod._arena()
        out_s, out_d = ops.planes_fwd(mod.layout, arena, xt_c, which)
        ctx.mod, ctx.which = mod, which
        ctx.save_for_backward(xt_c)
        ctx.need_dxt = xt.requires_grad
        outs = tuple(o for o in (out_s, out_d) if o is not None)
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *douts):
        (xt,) = ctx.saved_tensors
        mod, which = ctx.mod, ctx.which
        P = xt.shape[0]
        n_out = mod.layout.n_scales * mod.layout.C
        zeros = lambda: torch.zeros(P, n_out, dtype=torch.float32, device=xt.device)
        if which == 0:
            ds, dd = douts
        elif which == 1:
            ds, dd = douts[0], None
        else:
            ds, dd = None, douts[0]
        ds = None if which == 2 else (zeros() if ds is None else ds.float().contiguous())
        dd = None if which == 1 else (zeros() if dd is None else dd.float().contiguous())
        garena = torch.zeros(mod.layout.numel, dtype=torch.float32, device=xt.device)
        dxt = ops.planes_bwd(mod.layout, mod._arena(), xt, which, ds, dd, garena, ctx.need_dxt)
        grads = [torch.empty_like(p) for p in mod._flat_planes()]
        ops.planes_relayout(mod.layout, grads, garena, to_channel_last=False)
        return (dxt, None, None) + tuple(grads)


class Planes4D(nn.Module):
    def __init__(self, grid_dimensions=2, input_dim=4, output_dim=8, resolution=(32, 32, 32, 8),
                 multiscale_res=(1, 2, 4, 8), concat_ms_feat=True, decompose=True, reduction="prod"):
        super().__init__()
        if grid_dimensions != 2 or input_dim != 4 or not concat_ms_feat or not decompose or reduction != "prod":
            raise ValueError("Planes4D: only the configuration LiDAR4D uses is implemented "
                             "(2-D planes of a 4-D input, concat over scales, decompose, reduction='prod')")
        if output_dim != 8:
            raise ValueError("Planes4D: the HIP kernel is specialised for 8 channels per plane")
        self.config = {"grid_dimensions": grid_dimensions, "input_dim": input_dim, "output_dim": output_dim,
                       "resolution": list(resolution)}
        self.multiscale_res = list(multiscale_res)
        self.concat_ms_feat, self.decompose, self.reduction = concat_ms_feat, decompose, reduction
        combs = list(itertools.combinations(range(4), 2))
        self.planes = nn.ModuleList()
        res_per_scale = []
        for m in self.multiscale_res:
            reso = [r * m for r in resolution[:3]] + list(resolution[3:])  # multi-res only on spatial axes
            res_per_scale.append(reso)
            coefs = nn.ParameterList()
            for comb in combs:
                p = nn.Parameter(torch.empty([1, output_dim] + [reso[c] for c in comb[::-1]]))
                if 3 in comb:
                    nn.init.ones_(p)
                else:
                    nn.init.uniform_(p, a=0.1, b=0.5)
                coefs.append(p)
            self.planes.append(coefs)
        self.layout = ops.PlaneLayout(res_per_scale, output_dim)
        self.n_output_dims = output_dim * len(self.multiscale_res) * 2
        self._cl = None
        self._cl_key = None

    def _flat_planes(self):
        return [p for coefs in self.planes for p in coefs]

    def _arena(self):
        """Channel-last fp32 compute copy of all planes, refreshed when any plane changed."""
        planes = self._flat_planes()
        key = (planes[0].data_ptr(), _EPOCH[0], sum(p._version for p in planes))
        if key != self._cl_key:
            if self._cl is None or self._cl.device != planes[0].device:
                self._cl = torch.empty(self.layout.numel, dtype=torch.float32, device=planes[0].device)
            ops.planes_relayout(self.layout, [p.detach() for p in planes], self._cl, to_channel_last=True)
            self._cl_key = key
        return self._cl

    def forward_static(self, input):
        return _PlanesFn.apply(input, self, 1, *self._flat_planes())

    def forward_dynamic(self, input):
        return _PlanesFn.apply(input, self, 2, *self._flat_planes())

    def forward(self, input):
        s, d = _PlanesFn.apply(input, self, 0, *self._flat_planes())
        return [s, d]
