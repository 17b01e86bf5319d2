"""Volumetric LiDAR renderer.  Mirror of the reference's model/renderer.py:13-186 (``LiDAR_Renderer``):
same constructor, ``run`` / ``render`` signatures, result-dict keys and shapes, staged chunking.

``run`` is the fused HIP pipeline (lidar4d_amd/fused.py) when the subclass provides one (LiDAR4D does);
otherwise it composes the subclass's ``density`` / ``attribute`` with the sampling and compositing kernels
(l4d_sample_rays, l4d_composite_*), which is what a user-defined field subclass gets.
"""
import os

import torch
import torch.nn as nn

from . import ops


class _CompositeFn(torch.autograd.Function):
    """weights/depth/weights_sum from sigma (renderer.py:98-126) with the analytic adjoint."""

    @staticmethod
    def forward(ctx, sigma, z_vals, sample_dist, density_scale, active_sensor):
        sigma = sigma.detach().float().contiguous()
        weights, wsum, depth, mask, _, _ = ops.composite_fwd(sigma, z_vals, sample_dist, density_scale, active_sensor,
                                                            want_mask=True, want_idx=False)
        ctx.save_for_backward(sigma, z_vals, weights)
        ctx.cfg = (sample_dist, density_scale, active_sensor)
        ctx.mark_non_differentiable(mask)
        return weights, wsum, depth, mask

    @staticmethod
    def backward(ctx, d_weights, d_wsum, d_depth, _):
        sigma, z_vals, weights = ctx.saved_tensors
        sample_dist, density_scale, active_sensor = ctx.cfg
        c = lambda t: None if t is None else t.float().contiguous()
        d_sigma, _ = ops.composite_bwd(sigma, z_vals, weights, None, 0, sample_dist, density_scale, active_sensor,
                                       c(d_depth), c(d_wsum), None, c(d_weights), want_d_attr=False)
        return d_sigma, None, None, None, None


class LiDAR_Renderer(nn.Module):
    def __init__(self, bound=1, near_lidar=0.01, far_lidar=0.81, density_scale=1, active_sensor=False):
        super().__init__()
        self.bound = bound
        self.near_lidar = near_lidar
        self.far_lidar = far_lidar
        self.density_scale = density_scale
        self.active_sensor = active_sensor
        aabb = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb", aabb)
        self._lin_cache = {}

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def attribute(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def _lin(self, num_steps, device):
        key = (num_steps, str(device))
        if key not in self._lin_cache:
            self._lin_cache[key] = torch.linspace(0.0, 1.0, num_steps, device=device)  # renderer.py:77
        return self._lin_cache[key]

    def run(self, rays_o, rays_d, time, num_steps=768, perturb=False, noise=None, **kwargs):
        """Generic (un-fused) run for subclasses that only define density()/attribute()."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        device = rays_o.device
        if perturb and noise is None:
            noise = torch.rand(N, num_steps, device=device)  # renderer.py:84
        z_vals, xyzs = ops.sample_rays(rays_o, rays_d, self._lin(num_steps, device), noise if perturb else None,
                                       self.near_lidar, self.far_lidar, self.bound, want_xyz=True)
        dens = self.density(xyzs, time)
        sample_dist = (self.far_lidar - self.near_lidar) / num_steps
        weights, wsum, depth, mask = _CompositeFn.apply(dens["sigma"].reshape(N, num_steps), z_vals, sample_dist,
                                                        self.density_scale, self.active_sensor)
        dirs = rays_d.view(-1, 1, 3).expand(N, num_steps, 3).reshape(-1, 3)
        extra = {k: v for k, v in dens.items() if k != "sigma"}
        attr = self.attribute(xyzs, dirs, mask=mask.reshape(-1).bool(), **extra).view(N, num_steps, self.out_lidar_dim)
        image = torch.sum(weights.unsqueeze(-1) * attr, dim=-2)
        return {
            "depth_lidar": depth.view(*prefix),
            "image_lidar": image.view(*prefix, self.out_lidar_dim),
            "weights_sum_lidar": wsum,
            "weights": weights,
            "z_vals": z_vals,
        }

    # Staged inference renders a frame as 16-32 chunks of max_ray_batch rays (renderer.py:142-186).  With ``graph_staged``
    # (opt in: the attribute) full-size chunks of a no-grad render are captured ONCE into a hipGraph --
    # fixed shapes, the call's time stays on the device, no host decision depends on the data -- and replayed per chunk.
    # Bit-identical (tests/test_gpu_properties.py); measured on the 131,072-ray frame of BASELINE C5: 108.2 vs 107.7 ms, i.e.
    # nothing -- a 4096-ray chunk is 3.4 ms of kernel time and the host stays ahead of it.  Off by default; it pays where the
    # chunks are small (max_ray_batch of a few hundred rays).
    graph_staged = False

    def _run_chunk_graphed(self, rays_o, rays_d, time, kwargs):
        """One full-size chunk through the captured graph.  The graph is keyed on the parameter state (ParamStore._key):
        the host-side caches a forward refreshes when parameters changed (fp16 copies, slice-pair tables, channel-last
        planes) are refreshed by one eager chunk first, and any later change of the parameters drops the graph."""
        st = self.__dict__.setdefault("_chunk_graph", {"key": None, "seen": None})
        key = (self._store._key(), tuple(rays_o.shape), kwargs.get("num_steps", 768), str(rays_o.device))
        if st["key"] == key:
            st["rays_o"].copy_(rays_o), st["rays_d"].copy_(rays_d), st["time"].copy_(time)
            st["graph"].replay()
            return st["out"]
        if st["seen"] != key:
            st["seen"] = key
            return self.run(rays_o, rays_d, time, **kwargs)
        ro, rd, t = rays_o.clone(), rays_d.clone(), time.clone()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.run(ro, rd, t, **kwargs)
        st.update(key=key, graph=graph, rays_o=ro, rays_d=rd, time=t,
                  out={"depth_lidar": out["depth_lidar"], "image_lidar": out["image_lidar"]})
        graph.replay()  # capturing does not execute
        return st["out"]

    def render(self, rays_o, rays_d, time, staged=False, max_ray_batch=4096, **kwargs):
        _run = self.run
        B, N = rays_o.shape[:2]
        device = rays_o.device
        if staged:
            out_lidar_dim = self.out_lidar_dim
            depth = torch.empty((B, N), device=device)
            image = torch.empty((B, N, out_lidar_dim), device=device)
            graphed = (self.graph_staged and device.type == "cuda" and not torch.is_grad_enabled() and hasattr(self, "_store")
                       and not kwargs.get("perturb", False) and kwargs.get("noise") is None and torch.is_tensor(time)
                       and time.device == device)
            for b in range(B):
                head = 0
                while head < N:
                    tail = min(head + max_ray_batch, N)
                    if graphed and tail - head == max_ray_batch:
                        r = self._run_chunk_graphed(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], time[b:b + 1], kwargs)
                        depth[b:b + 1, head:tail] = r["depth_lidar"]
                        image[b:b + 1, head:tail] = r["image_lidar"]
                        head += max_ray_batch
                        continue
                    r = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], time[b:b + 1], **kwargs)
                    depth[b:b + 1, head:tail] = r["depth_lidar"]
                    image[b:b + 1, head:tail] = r["image_lidar"]
                    head += max_ray_batch
            return {"depth_lidar": depth, "image_lidar": image}
        return _run(rays_o, rays_d, time, **kwargs)
