"""LiDAR4D model glue.  Mirror of the reference's model/lidar4d.py:22-237: same constructor keyword arguments
(main_lidar4d.py:155-179), method names (``render``/``run``, ``density``, ``attribute``, ``flow``,
``get_params``) and state-dict keys, on HIP kernels.

* ``render`` / ``run`` -> the fused pipeline of lidar4d_amd/fused.py (what the reference's Trainer and
  Simulator call: runner.py:183,398,447; simulator.py:115).
* ``density`` / ``attribute`` / ``flow`` -> operator-level modules, for callers that use them directly
  (runner.py:227,252 calls ``flow``).

``self.unet`` is the ray-drop refinement network (lidar4d.py:119, lidar4d_amd/unet.py): dense convolutions through
MIOpen; it is not part of the flat parameter arenas (the reference trains it separately, runner.py:872).
"""
import numpy as np
import torch

from . import tcnn
from .activation import trunc_exp
from .flow_field import FlowField
from .fused import RenderFn
from .hash_field import HashGrid4D, _t_device
from .params import ParamStore
from .planes_field import Planes4D
from .renderer import LiDAR_Renderer
from .unet import UNet


class LiDAR4D(LiDAR_Renderer):
    def __init__(self, min_resolution=32, base_resolution=512, max_resolution=32768, time_resolution=8,
                 n_levels_plane=4, n_features_per_level_plane=8, n_levels_hash=8, n_features_per_level_hash=4,
                 log2_hashmap_size=19, num_layers_flow=3, hidden_dim_flow=64, num_layers_sigma=2, hidden_dim_sigma=64,
                 geo_feat_dim=15, num_layers_lidar=3, hidden_dim_lidar=64, out_lidar_dim=2, num_frames=51, bound=1,
                 **kwargs):
        super().__init__(bound, **kwargs)
        if out_lidar_dim != 2:
            raise ValueError("LiDAR4D: out_lidar_dim must be 2 (ray-drop, intensity) as in the reference")
        self.out_lidar_dim = out_lidar_dim
        self.num_frames = num_frames
        self.geo_feat_dim = geo_feat_dim
        self.loss_scale = 128.0  # scale of the fp16 adjoints inside the fused backward (tiny-cuda-nn's default)
        # True (for the reference's own loop: torch.optim.Adam + GradScaler, runner.py:474-551): after backward, the
        # HashGridT time slices the step did not use keep ``.grad = None`` as in the reference, so that torch's Adam skips
        # them (no moment decay, no step count).  Costs one host read of ``time`` per call -- the reference pays the same
        # in ``int(t * (num_frames - 1))`` (lidar4d.py:143).  lidar4d_amd.trainer.Trainer switches it off and gates those
        # ranges on the device instead.
        self.reference_grad_none = True

        self.planes_encoder = Planes4D(grid_dimensions=2, input_dim=4, output_dim=n_features_per_level_plane,
                                       resolution=[min_resolution] * 3 + [time_resolution],
                                       multiscale_res=[2 ** n for n in range(n_levels_plane)])
        self.hash_encoder = HashGrid4D(base_resolution=base_resolution, max_resolution=max_resolution,
                                       time_resolution=time_resolution, n_levels=n_levels_hash,
                                       n_features_per_level=n_features_per_level_hash,
                                       log2_hashmap_size=log2_hashmap_size)
        # the fused render pipeline (fused.py) is built for the reference model's own field layout -- four features per level,
        # blended with four time bases, the three 2-D x time stacks concatenated, static and dynamic part as a pair
        # (lidar4d.py:51-57 constructs nothing else); HashGrid4D's other options run on the operator-level path only
        he = self.hash_encoder
        if (n_features_per_level_hash != 4 or he.reduction != "concat" or not he.decompose
                or any(hd.num_basis != 4 for hd in he.hash_dynamic)):
            raise ValueError("LiDAR4D: the fused pipeline needs n_features_per_level_hash = 4 (num_basis 4), reduction = 'concat' and "
                             "decompose = True for its hash encoder; other HashGrid4D options are available at operator level only")
        self.view_encoder = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Frequency", "degree": 12})
        self.flow_net = FlowField(input_dim=4, num_layers=num_layers_flow, hidden_dim=hidden_dim_flow, use_grid=True)

        def net(n_in, n_out, hidden, layers):
            return tcnn.Network(n_input_dims=n_in, n_output_dims=n_out, network_config={
                "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                "n_neurons": hidden, "n_hidden_layers": layers - 1})

        self.sigma_net = net(self.planes_encoder.n_output_dims + self.hash_encoder.n_output_dims, 1 + geo_feat_dim,
                             hidden_dim_sigma, num_layers_sigma)
        self.intensity_net = net(self.view_encoder.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar)
        self.raydrop_net = net(self.view_encoder.n_output_dims + geo_feat_dim, 1, hidden_dim_lidar, num_layers_lidar)
        self.unet = UNet(in_channels=3, out_channels=1)  # lidar4d.py:119
        self._build_store()

    # -- flat parameter arenas -----------------------------------------------------------------------------
    def _build_store(self):
        named = lambda mod, prefix: [(prefix + n, p) for n, p in mod.named_parameters()]
        g0 = named(self.planes_encoder, "planes_encoder.") + named(self.hash_encoder, "hash_encoder.")
        g1 = (named(self.flow_net, "flow_net.") + named(self.sigma_net, "sigma_net.") +
              named(self.intensity_net, "intensity_net.") + named(self.raydrop_net, "raydrop_net."))
        last = "flow_net.mlp.%d.weight" % (len(self.flow_net.mlp) - 1)
        object.__setattr__(self, "_store", ParamStore([g0, g1], pad_to={last: 16 * 64}))
        for m in self.modules():
            if isinstance(m, (tcnn.HashGridEncoding, tcnn.FullyFusedMLP)):
                object.__setattr__(m, "_store", self._store)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._store.build()  # .to(device) / .half() replace every parameter's storage: re-flatten
        self._fd_key = None
        return out

    def zero_grad(self, set_to_none=True):
        if self._store.flat_grad is not None and not set_to_none:
            self._store.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)

    # -- fused path ------------------------------------------------------------------------------------------
    def run(self, rays_o, rays_d, time, num_steps=768, perturb=False, noise=None, time_host=None, **kwargs):
        """renderer.py:44-140.  ``noise`` ([N, num_steps] in [0,1)) replaces the internal torch.rand when given
        (parity tests feed the oracle's noise).  ``time_host``: the value of ``time`` as a python float when the caller
        has it (the training loop does): spares the one device read-back of a training step -- the host-side copy of
        hash_field.py:79-85's slice choice below, which would otherwise wait for the whole previous step to drain."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        device = rays_o.device
        if perturb and noise is None:
            noise = torch.rand(N, num_steps, device=device)
        if not perturb:
            noise = None
        elif noise is not None:
            noise = noise.to(device=device, dtype=torch.float32).contiguous()
        t_dev = _t_device(time, device)
        field_params = [p for _, p, _, n, _ in self._store.entries if n > 0]
        train = torch.is_grad_enabled() and any(p.requires_grad for p in field_params)
        params = [p for p in field_params if p.requires_grad] if train else []
        self._host_slice_pair = None
        if train and self.reference_grad_none:  # hash_field.py:79-85 on the host, fp32 like the reference's tensor math
            n_slices = self.hash_encoder.hash_dynamic[0].time_resolution
            t_host = time_host if time_host is not None else (float(time.reshape(-1)[0]) if torch.is_tensor(time) else float(time))
            idx = np.float32(t_host) * np.float32(n_slices - 1)
            self._host_slice_pair = (int(np.floor(idx)), int(np.ceil(idx)))
        depth, image, wsum, weights, z_vals, idx, count = RenderFn.apply(self, rays_o, rays_d, t_dev, noise, num_steps,
                                                                        train, *params)
        return {
            "depth_lidar": depth.view(*prefix),
            "image_lidar": image.view(*prefix, self.out_lidar_dim),
            "weights_sum_lidar": wsum,
            "weights": weights,
            "z_vals": z_vals,
            "mask_idx": idx,      # extra: compacted `weights > 1e-4` sample indices (renderer.py:110) ...
            "mask_count": count,  # ... and their number (device int32), the attribute work list
        }

    # -- operator-level API (lidar4d.py:124-223) -----------------------------------------------------------
    def flow(self, x, t):
        x = (x + self.bound) / (2 * self.bound)
        if t.shape[0] == 1:
            t = t.repeat(x.shape[0], 1)
        xt = torch.cat([x, t.to(x)], dim=-1)
        flow = self.flow_net(xt)
        return {"forward": flow[:, :3], "backward": flow[:, 3:]}

    def density(self, x, t=None):
        x = (x + self.bound) / (2 * self.bound)
        t = t.to(device=x.device, dtype=torch.float32).reshape(1, 1)
        frame_idx = int(np.float32(float(t)) * np.float32(self.num_frames - 1))
        hash_feat_s, hash_feat_d = self.hash_encoder(x, t)
        xt = torch.cat([x, t.expand(x.shape[0], 1)], dim=-1)
        plane_feat_s, plane_feat_d = self.planes_encoder(xt)
        flow = self.flow_net(xt).float()
        hash_feat_1 = hash_feat_2 = hash_feat_d
        plane_feat_1 = plane_feat_2 = plane_feat_d
        if frame_idx < self.num_frames - 1:
            x1 = x + flow[:, :3]
            t1 = torch.tensor((frame_idx + 1) / self.num_frames)
            with torch.no_grad():
                hash_feat_1 = self.hash_encoder.forward_dynamic(x1, t1)
            plane_feat_1 = self.planes_encoder.forward_dynamic(torch.cat([x1, t1.to(x1).expand(x1.shape[0], 1)], dim=-1))
        if frame_idx > 0:
            x2 = x + flow[:, 3:]
            t2 = torch.tensor((frame_idx - 1) / self.num_frames)
            with torch.no_grad():
                hash_feat_2 = self.hash_encoder.forward_dynamic(x2, t2)
            plane_feat_2 = self.planes_encoder.forward_dynamic(torch.cat([x2, t2.to(x2).expand(x2.shape[0], 1)], dim=-1))
        plane_feat_d = 0.5 * plane_feat_d + 0.25 * (plane_feat_1 + plane_feat_2)
        hash_feat_d = 0.5 * hash_feat_d + 0.25 * (hash_feat_1 + hash_feat_2)
        features = torch.cat([plane_feat_s, plane_feat_d, hash_feat_s.float(), hash_feat_d], dim=-1)
        h = self.sigma_net(features)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def attribute(self, x, d, mask=None, geo_feat=None, **kwargs):
        if mask is not None:
            output = torch.zeros(mask.shape[0], self.out_lidar_dim, dtype=torch.float32, device=x.device)
            if not mask.any():
                return output
            d = d[mask]
            geo_feat = geo_feat[mask]
        d = self.view_encoder((d + 1) / 2)
        inp = torch.cat([d, geo_feat.to(d.dtype)], dim=-1)
        intensity = torch.sigmoid(self.intensity_net(inp))
        raydrop = torch.sigmoid(self.raydrop_net(inp))
        h = torch.cat([raydrop, intensity], dim=-1)
        if mask is not None:
            output[mask] = h.to(output.dtype)
        else:
            output = h
        return output

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
