"""Oracle restatement of the reference's own torch code on the hot path.  TEST INFRASTRUCTURE ONLY.

Pinned against the real reference: ``oracle/make_golden.py`` drives the reference's modules
(imported from a scratch copy of /root/reference, build container only) on seeded inputs and
``tests/test_oracle_golden.py`` checks this file against the fixtures it wrote.

Every class keeps the reference's attribute names so a reference ``state_dict`` loads unchanged
(checkpoint row).  The arithmetic is written out explicitly (bilinear taps,
Lagrange coefficients, compositing recurrences) rather than through ``F.grid_sample`` & co so the
HIP kernels can be read against it line by line.
"""
import itertools

import numpy as np
import torch
import torch.nn as nn

from . import tcnn_ref as tcnn


# ----------------------------------------------------------------------------------------------
# shared: cubic Lagrange "interpT" (reference model/hash_field.py:65-74, model/flow_field.py:102-111)
# ----------------------------------------------------------------------------------------------
def lagrange_basis(t, num_basis=4):
    """Coefficients of the num_basis Lagrange polynomials with nodes i/(num_basis-1) at ``t``.
    ``t`` is a tensor (any shape with one element); arithmetic follows the reference's
    ``(t - T[m]) / (T[j] - T[m])`` product order so fp32 results agree to the last bit."""
    T = [i / (num_basis - 1) for i in range(num_basis)]
    coefs = []
    for j in range(num_basis):
        c = 1
        for m in range(num_basis):
            if m != j:
                c = c * ((t - T[m]) / (T[j] - T[m]))
        coefs.append(c)
    return coefs


def interp_t(feat, t, n_levels, n_features, num_basis=4):
    """[P, L*F] -> [P, L*F/num_basis]: split each level's F features into num_basis chunks and
    combine them with the Lagrange basis at ``t``."""
    x = feat.view(-1, n_levels, n_features)
    chunks = torch.chunk(x, num_basis, dim=-1)
    coefs = lagrange_basis(t, num_basis)
    acc = 0
    for c, ch in zip(coefs, chunks):
        acc = acc + c * ch
    return acc.reshape(feat.shape[0], n_levels * n_features // num_basis)


# ----------------------------------------------------------------------------------------------
# hash field (reference model/hash_field.py)
# ----------------------------------------------------------------------------------------------
class HashGridT(nn.Module):
    """2-D hash grids at ``time_resolution`` time slices, linearly blended in time, then interpT.
    Reference: model/hash_field.py:30-88."""

    def __init__(self, time_resolution=8, base_resolution=512, max_resolution=32768, n_levels=8,
                 n_features_per_level=4, log2_hashmap_size=14, num_basis=4):
        super().__init__()
        self.time_resolution = time_resolution
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        cfg = {
            "otype": "HashGrid",
            "n_levels": n_levels,
            "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size,
            "base_resolution": base_resolution,
            "per_level_scale": per_level_scale,
        }
        self.hash_t = nn.ModuleList([tcnn.Encoding(2, cfg) for _ in range(time_resolution)])
        self.n_levels = n_levels
        self.n_features_per_level = n_features_per_level
        self.num_basis = num_basis
        self.n_output_dims = n_levels * n_features_per_level // num_basis

    def forward(self, x, t):
        # reference hash_field.py:79-86 -- slice pair + linear weights, all fp32 tensor arithmetic
        t = torch.as_tensor(t, dtype=torch.float32)
        idx = t * (self.time_resolution - 1)
        i1 = int(torch.floor(idx))
        i2 = int(torch.ceil(idx))
        if i1 == i2:
            feat = self.hash_t[i1](x).float()
        else:
            feat = (i2 - idx) * self.hash_t[i1](x).float() + (idx - i1) * self.hash_t[i2](x).float()
        return interp_t(feat, t, self.n_levels, self.n_features_per_level, self.num_basis)


class HashGrid4D(nn.Module):
    """Static 3-D hash grid + xy/xz/yz HashGridT stacks.  Reference: model/hash_field.py:91-172."""

    def __init__(self, base_resolution=512, max_resolution=32768, time_resolution=8, n_levels=8,
                 n_features_per_level=4, log2_hashmap_size=19, hash_size_dynamic=(15, 13, 13), decompose=True, reduction="concat"):
        super().__init__()
        self.decompose, self.reduction = decompose, reduction
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        self.hash_static = tcnn.Encoding(3, {
            "otype": "HashGrid",
            "n_levels": n_levels,
            "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size,
            "base_resolution": base_resolution,
            "per_level_scale": per_level_scale,
        })
        self.hash_dynamic = nn.ModuleList([
            HashGridT(time_resolution, base_resolution, max_resolution, n_levels,
                      n_features_per_level, hash_size_dynamic[i]) for i in range(3)
        ])
        n_dyn = self.hash_dynamic[0].n_output_dims
        self.n_output_dims = self.hash_static.n_output_dims + (3 * n_dyn if reduction == "concat" else n_dyn)  # hash_field.py:134-138

    def forward_static(self, x):
        return self.hash_static(x).float()

    def forward_dynamic(self, x, t):
        pairs = ((0, 1), (0, 2), (1, 2))  # xy, xz, yz  (hash_field.py:147-153)
        feats = [self.hash_dynamic[i](x[:, list(p)], t) for i, p in enumerate(pairs)]
        if self.reduction == "concat":  # hash_field.py:16-27 (reduction_func), :155-156
            return torch.cat(feats, -1)
        if self.reduction == "prod":
            return feats[0] * feats[1] * feats[2]
        if self.reduction == "sum":
            return feats[0] + feats[1] + feats[2]
        if self.reduction == "mean":
            return (feats[0] + feats[1] + feats[2]) / 3
        raise ValueError("Invalid reduction")

    def forward(self, x, t):
        static, dynamic = self.forward_static(x), self.forward_dynamic(x, t)
        return [static, dynamic] if self.decompose else torch.cat([static, dynamic], dim=-1)  # hash_field.py:164-170


# ----------------------------------------------------------------------------------------------
# hex-planes (reference model/planes_field.py)
# ----------------------------------------------------------------------------------------------
def bilinear_border(plane, cx, cy):
    """Bilinear sample of ``plane`` [1, C, H, W] at normalised coords cx (-> W axis), cy (-> H axis)
    in [0,1], align_corners=True, border padding: what ``grid_sample_wrapper``
    (planes_field.py:56-84) asks ATen for.  Returns [P, C].  Gradients wrt cx/cy follow ATen's
    rule: zero when the un-normalised coordinate was clipped (<=0 or >=size-1)."""
    _, C, H, W = plane.shape

    def unnorm(c, size):
        p = ((c * 2.0 - 1) + 1.0) / 2 * (size - 1)
        inside = (p > 0) & (p < size - 1)
        pc = p.clamp(0, size - 1)
        return torch.where(inside, pc, pc.detach())

    ix, iy = unnorm(cx, W), unnorm(cy, H)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    wx1, wy1 = ix - x0, iy - y0
    wx0, wy0 = (x0 + 1) - ix, (y0 + 1) - iy
    x0i, y0i = x0.long(), y0.long()
    x1i, y1i = (x0i + 1).clamp(max=W - 1), (y0i + 1).clamp(max=H - 1)  # weight is 0 when clamped
    g = plane[0].permute(1, 2, 0)  # [H, W, C]
    out = (wx0 * wy0).unsqueeze(-1) * g[y0i, x0i]
    out = out + (wx1 * wy0).unsqueeze(-1) * g[y0i, x1i]
    out = out + (wx0 * wy1).unsqueeze(-1) * g[y1i, x0i]
    out = out + (wx1 * wy1).unsqueeze(-1) * g[y1i, x1i]
    return out


PLANE_COMBS = list(itertools.combinations(range(4), 2))  # (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)


class Planes4D(nn.Module):
    """K-Planes style hex-plane field.  Reference: model/planes_field.py:144-239 with the defaults
    LiDAR4D passes (lidar4d.py:51-57): grid_dimensions=2, reduction='prod', concat over scales."""

    def __init__(self, output_dim=8, resolution=(32, 32, 32, 8), multiscale_res=(1, 2, 4, 8)):
        super().__init__()
        self.multiscale_res = list(multiscale_res)
        self.planes = nn.ModuleList()
        for m in self.multiscale_res:
            reso = [r * m for r in resolution[:3]] + list(resolution[3:])
            coefs = nn.ParameterList()
            for comb in PLANE_COMBS:
                p = nn.Parameter(torch.empty([1, output_dim] + [reso[c] for c in comb[::-1]]))
                if 3 in comb:
                    nn.init.ones_(p)
                else:
                    nn.init.uniform_(p, a=0.1, b=0.5)
                coefs.append(p)
            self.planes.append(coefs)
        self.n_output_dims = output_dim * len(self.multiscale_res) * 2

    def _sample(self, xt, which):
        stat, dyn = [], []
        for coefs in self.planes:
            fs, fd = None, None
            for ci, comb in enumerate(PLANE_COMBS):
                is_t = 3 in comb
                if (which == "static" and is_t) or (which == "dynamic" and not is_t):
                    continue
                v = bilinear_border(coefs[ci], xt[:, comb[0]], xt[:, comb[1]])
                if is_t:
                    fd = v if fd is None else fd * v
                else:
                    fs = v if fs is None else fs * v
            stat.append(fs)
            dyn.append(fd)
        s = torch.cat(stat, -1) if which != "dynamic" else None
        d = torch.cat(dyn, -1) if which != "static" else None
        return s, d

    def forward_static(self, xt):
        return self._sample(xt, "static")[0]

    def forward_dynamic(self, xt):
        return self._sample(xt, "dynamic")[1]

    def forward(self, xt):
        return list(self._sample(xt, "both"))


# ----------------------------------------------------------------------------------------------
# flow field (reference model/flow_field.py)
# ----------------------------------------------------------------------------------------------
class FlowField(nn.Module):
    """3-D hash grid (8 lv x 8 feat) -> interpT -> bias-free ReLU MLP 16->64->64->6.
    Reference: model/flow_field.py:40-130 (use_freq=False, use_grid=True).

    In tcnn precision mode the three ``nn.Linear`` layers follow the autocast-fp16 contract the
    reference trains under (runner.py:497): fp16 operands and activations, fp32 accumulation."""

    def __init__(self, num_layers=3, hidden_dim=64, num_basis=4, n_levels=8, n_features_per_level=8,
                 base_resolution=32, max_resolution=8192, log2_hashmap_size=18):
        super().__init__()
        per_level_scale = np.exp2(np.log2(max_resolution / base_resolution) / (n_levels - 1))
        self.grid_enc = tcnn.Encoding(3, {
            "otype": "HashGrid",
            "n_levels": n_levels,
            "n_features_per_level": n_features_per_level,
            "log2_hashmap_size": log2_hashmap_size,
            "base_resolution": base_resolution,
            "per_level_scale": per_level_scale,
        })
        self.n_levels, self.n_features_per_level, self.num_basis = n_levels, n_features_per_level, num_basis
        self.input_dim = self.grid_enc.n_output_dims // num_basis
        layers = []
        for l in range(num_layers):
            i = self.input_dim if l == 0 else hidden_dim
            o = 6 if l == num_layers - 1 else hidden_dim
            layers.append(nn.Linear(i, o, bias=False))
            if l != num_layers - 1:
                layers.append(nn.ReLU())
        self.mlp = nn.Sequential(*layers)
        nn.init.normal_(self.mlp[-1].weight.data, 0, 0.001)

    def forward(self, xt):
        t = xt[0, 3]
        g = self.grid_enc(xt[:, :3]).float()
        h = tcnn.rh(interp_t(g, t, self.n_levels, self.n_features_per_level, self.num_basis))
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        for m in lin[:-1]:
            h = tcnn.rh(torch.relu(h @ tcnn.rh(m.weight).t()))
        return tcnn.rh(h @ tcnn.rh(lin[-1].weight).t())


# ----------------------------------------------------------------------------------------------
# trunc_exp (reference model/activation.py:6-20)
# ----------------------------------------------------------------------------------------------
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


# ----------------------------------------------------------------------------------------------
# renderer + model glue (reference model/renderer.py, model/lidar4d.py)
# ----------------------------------------------------------------------------------------------
def sample_z(n_rays, near, far, num_steps, noise=None):
    """z_vals [N, T] of renderer.py:77-85; ``noise`` stands in for ``torch.rand(z_vals.shape)``."""
    z = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand(n_rays, num_steps)
    nears = torch.ones(n_rays, 1) * near
    fars = torch.ones(n_rays, 1) * far
    z = nears + (fars - nears) * z
    sample_dist = (fars - nears) / num_steps
    if noise is not None:
        z = z + (noise - 0.5) * sample_dist
    return z, sample_dist


def composite(sigma, z_vals, sample_dist, density_scale=1.0, active_sensor=False):
    """weights [N, T] of renderer.py:98-104."""
    deltas = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], sample_dist * torch.ones_like(z_vals[:, :1])], -1)
    k = 2.0 if active_sensor else 1.0
    alphas = 1 - torch.exp(-k * deltas * density_scale * sigma)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-15], -1)
    return alphas * torch.cumprod(shifted, -1)[:, :-1]


class LiDAR4D(nn.Module):
    """Reference model/lidar4d.py:22-237 on top of model/renderer.py:13-186 (one class here)."""

    def __init__(self, min_resolution=32, base_resolution=512, max_resolution=32768, time_resolution=8,
                 n_levels_plane=4, n_features_per_level_plane=8, n_levels_hash=8,
                 n_features_per_level_hash=4, log2_hashmap_size=19, num_layers_flow=3,
                 hidden_dim_flow=64, num_layers_sigma=2, hidden_dim_sigma=64, geo_feat_dim=15,
                 num_layers_lidar=3, hidden_dim_lidar=64, out_lidar_dim=2, num_frames=51, bound=1,
                 near_lidar=0.01, far_lidar=0.81, density_scale=1, active_sensor=False):
        super().__init__()
        self.bound, self.near_lidar, self.far_lidar = bound, near_lidar, far_lidar
        self.density_scale, self.active_sensor = density_scale, active_sensor
        self.register_buffer("aabb", torch.FloatTensor([-bound] * 3 + [bound] * 3))
        self.out_lidar_dim, self.num_frames = out_lidar_dim, num_frames
        self.planes_encoder = Planes4D(n_features_per_level_plane,
                                       [min_resolution] * 3 + [time_resolution],
                                       [2 ** n for n in range(n_levels_plane)])
        self.hash_encoder = HashGrid4D(base_resolution, max_resolution, time_resolution, n_levels_hash,
                                       n_features_per_level_hash, log2_hashmap_size)
        self.view_encoder = tcnn.Encoding(3, {"otype": "Frequency", "degree": 12})
        self.flow_net = FlowField(num_layers=num_layers_flow, hidden_dim=hidden_dim_flow)
        mlp = lambda i, o, n, l: tcnn.Network(i, o, {
            "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
            "n_neurons": n, "n_hidden_layers": l - 1})
        self.sigma_net = mlp(self.planes_encoder.n_output_dims + self.hash_encoder.n_output_dims,
                             1 + geo_feat_dim, hidden_dim_sigma, num_layers_sigma)
        self.intensity_net = mlp(self.view_encoder.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar)
        self.raydrop_net = mlp(self.view_encoder.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar)

    # -- lidar4d.py:124-137
    def flow(self, x, t):
        x = (x + self.bound) / (2 * self.bound)
        xt = torch.cat([x, t.reshape(1, 1).expand(x.shape[0], 1)], -1)
        f = self.flow_net(xt).float()
        return {"forward": f[:, :3], "backward": f[:, 3:]}

    # -- lidar4d.py:139-188
    def density(self, x, t):
        x = (x + self.bound) / (2 * self.bound)
        t = t.reshape(1, 1).float()
        frame_idx = int(t * (self.num_frames - 1))
        hash_s, hash_d = self.hash_encoder(x, t)
        xt = torch.cat([x, t.expand(x.shape[0], 1)], -1)
        plane_s, plane_d = self.planes_encoder(xt)
        flow = self.flow_net(xt).float()
        hash_1 = hash_2 = hash_d
        plane_1 = plane_2 = plane_d
        if frame_idx < self.num_frames - 1:
            x1 = x + flow[:, :3]
            t1 = torch.tensor((frame_idx + 1) / self.num_frames)  # /num_frames quirk, lidar4d.py:159
            with torch.no_grad():
                hash_1 = self.hash_encoder.forward_dynamic(x1, t1)
            plane_1 = self.planes_encoder.forward_dynamic(torch.cat([x1, t1.expand(x.shape[0], 1)], -1))
        if frame_idx > 0:
            x2 = x + flow[:, 3:]
            t2 = torch.tensor((frame_idx - 1) / self.num_frames)
            with torch.no_grad():
                hash_2 = self.hash_encoder.forward_dynamic(x2, t2)
            plane_2 = self.planes_encoder.forward_dynamic(torch.cat([x2, t2.expand(x.shape[0], 1)], -1))
        plane_d = 0.5 * plane_d + 0.25 * (plane_1 + plane_2)
        hash_d = 0.5 * hash_d + 0.25 * (hash_1 + hash_2)
        feats = torch.cat([plane_s, plane_d, hash_s, hash_d], -1)
        h = self.sigma_net(feats)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    # -- lidar4d.py:191-223
    def attribute(self, x, d, mask=None, geo_feat=None):
        out = torch.zeros(x.shape[0], self.out_lidar_dim, dtype=torch.float32)
        if mask is not None:
            if not mask.any():
                return out
            d, geo_feat = d[mask], geo_feat[mask]
        d = self.view_encoder((d + 1) / 2)
        inp = torch.cat([d.float(), geo_feat.float()], -1)
        # sigmoid on the network's fp16 output stays fp16 in the reference (then cast to fp32)
        intensity = tcnn.rh(torch.sigmoid(self.intensity_net(inp).float()))
        raydrop = tcnn.rh(torch.sigmoid(self.raydrop_net(inp).float()))
        h = torch.cat([raydrop, intensity], -1)
        if mask is None:
            return h
        out = out.clone()
        out[mask] = h
        return out

    # -- renderer.py:44-140
    def run(self, rays_o, rays_d, time, num_steps=768, perturb=False, noise=None, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        if perturb and noise is None:
            noise = torch.rand(N, num_steps)
        z_vals, sample_dist = sample_z(N, self.near_lidar, self.far_lidar, num_steps, noise if perturb else None)
        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, self.aabb[:3]), self.aabb[3:])
        dens = self.density(xyzs.reshape(-1, 3), time)
        sigma = dens["sigma"].view(N, num_steps)
        weights = composite(sigma, z_vals, sample_dist, self.density_scale, self.active_sensor)
        mask = weights > 1e-4
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        attr = self.attribute(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1),
                              geo_feat=dens["geo_feat"]).view(N, num_steps, self.out_lidar_dim)
        return {
            "depth_lidar": (weights * z_vals).sum(-1).view(*prefix),
            "image_lidar": (weights.unsqueeze(-1) * attr).sum(-2).view(*prefix, self.out_lidar_dim),
            "weights_sum_lidar": weights.sum(-1),
            "weights": weights,
            "z_vals": z_vals,
            "mask": mask,  # oracle extra: the weights>1e-4 index set (renderer.py:110)
        }

    # -- renderer.py:142-186
    def render(self, rays_o, rays_d, time, staged=False, max_ray_batch=4096, **kwargs):
        if not staged:
            return self.run(rays_o, rays_d, time, **kwargs)
        B, N = rays_o.shape[:2]
        depth = torch.empty(B, N)
        image = torch.empty(B, N, self.out_lidar_dim)
        for b in range(B):
            for head in range(0, N, max_ray_batch):
                tail = min(head + max_ray_batch, N)
                r = self.run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], time[b:b + 1], **kwargs)
                depth[b:b + 1, head:tail] = r["depth_lidar"]
                image[b:b + 1, head:tail] = r["image_lidar"]
        return {"depth_lidar": depth, "image_lidar": image}

    # -- lidar4d.py:226-237
    def get_params(self, lr):
        return [
            {"params": self.planes_encoder.parameters(), "lr": lr},
            {"params": self.hash_encoder.parameters(), "lr": lr},
            {"params": self.view_encoder.parameters(), "lr": lr},
            {"params": self.flow_net.parameters(), "lr": 0.1 * lr},
            {"params": self.sigma_net.parameters(), "lr": 0.1 * lr},
            {"params": self.intensity_net.parameters(), "lr": 0.1 * lr},
            {"params": self.raydrop_net.parameters(), "lr": 0.1 * lr},
        ]
