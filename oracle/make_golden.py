"""Generate tests/golden/*.npz by running the REAL reference modules.  Build container only.

    python -m oracle.make_golden            # needs /root/reference; writes tests/golden/

The reference's python sources are imported from a throw-away scratch copy (importing
utils/chamfer3D would otherwise write build artefacts next to the sources); nothing of
the reference is copied into this repo -- the fixtures hold inputs, seeds, configs and expected
outputs only.  tiny-cuda-nn is not installable here, so the module name ``tinycudann`` is bound to
``oracle.tcnn_ref`` in **fp32 mode** while the reference's glue runs: the fixtures therefore pin
everything the reference itself implements (renderer, hash_field glue, planes_field, flow_field
glue, lidar4d, activation, get_lidar_rays) and leave the tiny-cuda-nn operators "parity
unpinned" (oracle/__init__.py).
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

from oracle import tcnn_ref
from oracle.detparams import det_uniform, fill_model, grad_digest

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    scratch = tempfile.mkdtemp(prefix="l4d_ref_")
    dst = os.path.join(scratch, "ref")
    shutil.copytree(REF, dst)
    sys.modules["tinycudann"] = tcnn_ref
    sys.path.insert(0, dst)
    import model.lidar4d as ref_lidar4d  # noqa
    import model.renderer as ref_renderer  # noqa
    import model.planes_field as ref_planes  # noqa
    import model.hash_field as ref_hash  # noqa
    import model.flow_field as ref_flow  # noqa
    import model.activation as ref_act  # noqa
    import data.base_dataset as ref_data  # noqa
    return dict(lidar4d=ref_lidar4d, renderer=ref_renderer, planes=ref_planes, hash=ref_hash,
                flow=ref_flow, act=ref_act, data=ref_data, scratch=scratch)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


SMALL_MODEL = dict(min_resolution=16, base_resolution=512, max_resolution=32768, time_resolution=8,
                   n_levels_plane=2, n_features_per_level_plane=8, n_levels_hash=4,
                   n_features_per_level_hash=4, log2_hashmap_size=14, num_layers_flow=3,
                   hidden_dim_flow=64, num_layers_sigma=2, hidden_dim_sigma=64, geo_feat_dim=15,
                   num_layers_lidar=3, hidden_dim_lidar=64, out_lidar_dim=2, num_frames=51, bound=1,
                   near_lidar=1.0 * 0.010504329815187737, far_lidar=81.0 * 0.010504329815187737,
                   density_scale=1, active_sensor=False)


def test_rays(n, seed):
    """n deterministic KITTI-360-shaped rays: origin near the scene centre, unit directions."""
    o = det_uniform((1, 3), f"o{seed}", -0.05, 0.05).expand(n, 3).contiguous()
    az = det_uniform((n,), f"az{seed}", -np.pi, np.pi)
    el = det_uniform((n,), f"el{seed}", np.deg2rad(-24.9), np.deg2rad(2.0))
    d = torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], -1)
    return o.unsqueeze(0), d.unsqueeze(0)


def main():
    torch.set_num_threads(8)
    tcnn_ref.set_precision("fp32")
    R = _import_reference()

    # (1) get_lidar_rays, 64 x 1024, identity and a rotated+translated pose
    H, W = 64, 1024
    c, s = np.cos(0.3), np.sin(0.3)
    poses = torch.tensor([np.eye(4), [[c, -s, 0, 0.1], [s, c, 0, -0.2], [0, 0, 1, 0.03], [0, 0, 0, 1]]], dtype=torch.float32)
    r = R["data"].get_lidar_rays(poses, [2.0, 26.9], H, W, -1)
    sel = torch.arange(0, H * W, 61)
    save("rays_64x1024", poses=poses, fov=np.array([2.0, 26.9]), H=H, W=W, sel=sel,
         rays_o=r["rays_o"][:, sel], rays_d=r["rays_d"][:, sel],
         rays_d_sum=r["rays_d"].double().sum(1), rays_d_abs_sum=r["rays_d"].double().abs().sum(1))

    # (2) LiDAR_Renderer.run with an analytic density / attribute (pins R1-R3 only)
    class Analytic(R["renderer"].LiDAR_Renderer):
        out_lidar_dim = 2

        def density(self, x, t):
            r2 = ((x - torch.tensor([0.2, 0.1, -0.05])) ** 2).sum(-1)
            sig = 400.0 * torch.exp(-r2 / 0.02) + 0.3
            return {"sigma": sig, "geo_feat": torch.stack([x[:, 0], x[:, 1] * 2], -1)}

        def attribute(self, x, d, mask=None, geo_feat=None, **kw):
            out = torch.zeros(x.shape[0], 2)
            a = torch.stack([torch.sigmoid(geo_feat[:, 0] * 3 + d[:, 0]), torch.sigmoid(geo_feat[:, 1] - d[:, 2])], -1)
            out[mask] = a[mask]
            return out

    ro, rd = test_rays(48, 1)
    time = torch.tensor([[0.3]])
    for tag, active, perturb, bound in (("plain", False, False, 1), ("active_perturb", True, True, 1), ("tightbound", False, True, 0.3)):
        ren = Analytic(bound=bound, near_lidar=SMALL_MODEL["near_lidar"], far_lidar=SMALL_MODEL["far_lidar"],
                       density_scale=1.5, active_sensor=active)
        noise = det_uniform((48, 768), "noise" + tag, 0.0, 1.0)
        orig_rand = torch.rand
        torch.rand = lambda *a, **k: noise  # capture the reference's torch.rand(z_vals.shape)
        try:
            out = ren.run(ro, rd, time, num_steps=768, perturb=perturb)
        finally:
            torch.rand = orig_rand
        save(f"run_analytic_{tag}", rays_o=ro, rays_d=rd, noise=noise, perturb=perturb, active=active, bound=bound,
             density_scale=1.5, near=SMALL_MODEL["near_lidar"], far=SMALL_MODEL["far_lidar"],
             z_vals=out["z_vals"], weights=out["weights"], depth=out["depth_lidar"], image=out["image_lidar"],
             weights_sum=out["weights_sum_lidar"], mask_idx=torch.nonzero((out["weights"] > 1e-4).reshape(-1)).reshape(-1))

    # (3) Planes4D forward / static / dynamic + grads wrt planes and coords
    pl = R["planes"].Planes4D(grid_dimensions=2, input_dim=4, output_dim=8, resolution=[8, 8, 8, 8], multiscale_res=[1, 2, 4])
    with torch.no_grad():
        for n, p in pl.named_parameters():
            ci = n.split(".")[-1]
            lo, hi = (0.8, 1.2) if ci in ("2", "4", "5") else (0.1, 0.5)
            p.copy_(det_uniform(tuple(p.shape), "pl:" + n, lo, hi))
    xt = det_uniform((2048, 4), "plx", -0.05, 1.05)
    xt[:64] = xt[:64].clamp(0, 1).round()  # exact border hits
    xt.requires_grad_(True)
    fs, fd = pl(xt)
    gs, gd = det_uniform(tuple(fs.shape), "gs", -1, 1), det_uniform(tuple(fd.shape), "gd", -1, 1)
    ((fs * gs).sum() + (fd * gd).sum()).backward()
    arrays = {f"param.{n}": p for n, p in pl.named_parameters()}
    arrays.update({f"grad.{n}": p.grad for n, p in pl.named_parameters()})
    save("planes4d", xt=xt, feat_static=fs, feat_dynamic=fd, gs=gs, gd=gd, grad_xt=xt.grad,
         feat_static_only=pl.forward_static(xt), feat_dynamic_only=pl.forward_dynamic(xt), **arrays)

    # (4) HashGridT / HashGrid4D glue (time blend + interpT) at t in {0, 0.3, 1} and a 0-dim t
    hg = R["hash"].HashGrid4D(base_resolution=16, max_resolution=256, time_resolution=8, n_levels=4,
                              n_features_per_level=4, log2_hashmap_size=10, hash_size_dynamic=[8, 7, 7])
    with torch.no_grad():
        for n, p in hg.named_parameters():
            p.copy_(det_uniform(tuple(p.shape), "hg:" + n, -0.5, 0.5))
    x = det_uniform((1024, 3), "hgx", 0.0, 1.0)
    arrays = {}
    for t in (0.0, 0.3, 1.0):
        s_, d_ = hg(x, torch.tensor([[t]]))
        arrays[f"static_t{t}"], arrays[f"dynamic_t{t}"] = s_, d_
    arrays["dynamic_t0dim_26_51"] = hg.forward_dynamic(x, torch.tensor(26 / 51))
    d_ = hg.forward_dynamic(x, torch.tensor([[0.62]]))
    gd = det_uniform(tuple(d_.shape), "hgd", -1, 1)
    (d_ * gd).sum().backward()
    arrays.update({f"grad.{n}": p.grad for n, p in hg.named_parameters() if p.grad is not None})
    save("hashgrid4d_glue", x=x, gd=gd, dynamic_t062=d_, **arrays)

    # (5)-(7) full model: density / attribute / render forward + backward
    model = R["lidar4d"].LiDAR4D(**SMALL_MODEL)
    fill_model(model, seed=7)
    pts = det_uniform((512, 3), "dpts", -1.0, 1.0)
    arrays = {}
    for fi in (0, 25, 50):
        model.zero_grad()
        out = model.density(pts, torch.tensor([[fi / 50]]))
        gsig = det_uniform((512,), f"gsig{fi}", -1, 1)
        ggeo = det_uniform((512, 15), f"ggeo{fi}", -1, 1)
        ((out["sigma"] * gsig).sum() + (out["geo_feat"] * ggeo).sum()).backward()
        arrays[f"sigma_f{fi}"], arrays[f"geo_f{fi}"] = out["sigma"], out["geo_feat"]
        for n, v in grad_digest(model).items():
            arrays[f"gdig_f{fi}.{n}"] = v
        fl = model.flow(pts, torch.tensor([[fi / 50]]))
        arrays[f"flow_fwd_f{fi}"], arrays[f"flow_bwd_f{fi}"] = fl["forward"], fl["backward"]
    save("density_small", pts=pts, cfg_seed=7, **arrays)

    d_in = torch.nn.functional.normalize(det_uniform((512, 3), "adir", -1, 1), dim=-1)
    geo = det_uniform((512, 15), "ageo", -1, 1)
    arrays = {}
    for tag, m in (("empty", torch.zeros(512, dtype=torch.bool)), ("sparse", det_uniform((512,), "am", 0, 1) > 0.8),
                   ("full", torch.ones(512, dtype=torch.bool))):
        arrays[f"mask_{tag}"] = m
        arrays[f"out_{tag}"] = model.attribute(pts, d_in, mask=m, geo_feat=geo)
    save("attribute_small", pts=pts, dirs=d_in, geo=geo, **arrays)

    for tag, frame, n_rays, steps, dscale, perturb in (("f25_T96", 25, 64, 96, 40.0, True), ("f0_T768", 0, 16, 768, 8.0, False),
                                                        ("f50_T64", 50, 32, 64, 100.0, True)):
        model.density_scale = dscale
        model.zero_grad()
        ro, rd = test_rays(n_rays, 11 + frame)
        noise = det_uniform((n_rays, steps), "rnoise" + tag, 0.0, 1.0)
        orig_rand = torch.rand
        torch.rand = lambda *a, **k: noise
        try:
            out = model.render(ro, rd, torch.tensor([[frame / 50]]), staged=False, num_steps=steps, perturb=perturb)
        finally:
            torch.rand = orig_rand
        gd_ = det_uniform(tuple(out["depth_lidar"].shape), "gdep" + tag, -1, 1)
        gi_ = det_uniform(tuple(out["image_lidar"].shape), "gimg" + tag, -1, 1)
        ((out["depth_lidar"] * gd_).sum() + (out["image_lidar"] * gi_).sum()).backward()
        arrays = {f"gdig.{n}": v for n, v in grad_digest(model).items()}
        save(f"render_small_{tag}", rays_o=ro, rays_d=rd, noise=noise, frame=frame, num_steps=steps, density_scale=dscale,
             perturb=perturb, depth=out["depth_lidar"], image=out["image_lidar"], weights_sum=out["weights_sum_lidar"],
             weights=out["weights"], z_vals=out["z_vals"], gdep=gd_, gimg=gi_,
             mask_idx=torch.nonzero((out["weights"] > 1e-4).reshape(-1)).reshape(-1), **arrays)
        # staged path == chunked run (renderer.py:159-177)
        with torch.no_grad():
            st = model.render(ro, rd, torch.tensor([[frame / 50]]), staged=True, max_ray_batch=24, num_steps=steps, perturb=False)
        save(f"render_small_{tag}_staged", depth=st["depth_lidar"], image=st["image_lidar"])

    shutil.rmtree(R["scratch"], ignore_errors=True)


if __name__ == "__main__":
    main()
