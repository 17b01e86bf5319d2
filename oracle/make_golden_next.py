"""Fixtures for the SURVEY 8f "next" rows (U-Net refinement, range-image <-> point-cloud conversions), produced by
running the REAL reference code.  Build container only:

    python -m oracle.make_golden_next       # needs /root/reference; writes tests/golden/{unet_eval,convert}.npz

Same policy as make_golden.py: the reference is imported from a throw-away scratch copy, only inputs / expected outputs /
key lists are saved.  TEST INFRASTRUCTURE.
"""
import shutil

import numpy as np
import torch

from oracle.detparams import convert_inputs, det_uniform, fill_unet
from oracle.make_golden import _import_reference, save


def gen_unet():
    import model.unet as ref_unet  # from the scratch copy (sys.path set by _import_reference)

    net = fill_unet(ref_unet.UNet(in_channels=3, out_channels=1), seed=0).eval()
    keys = list(net.state_dict().keys())
    shapes = [list(v.shape) for v in net.state_dict().values()]
    out = {}
    for tag, (b, h, w) in {"a": (2, 32, 96), "b": (1, 34, 70)}.items():  # "b": odd sizes exercise the centre padding
        x = det_uniform((b, 3, h, w), "unet_in_" + tag, 0.0, 1.0).requires_grad_(True)
        y = net(x)
        gy = det_uniform(tuple(y.shape), "unet_gy_" + tag, -1.0, 1.0)
        net.zero_grad()
        (y * gy).sum().backward()
        out["x_" + tag], out["y_" + tag], out["gy_" + tag], out["gx_" + tag] = x.detach(), y.detach(), gy, x.grad
        out["gw_inc_" + tag] = net.inc.conv.weight.grad.clone()
        g = net.attn.proj_qkv.weight.grad.double()
        out["gw_attn_digest_" + tag] = np.array([g.sum().item(), g.abs().sum().item()])
    save("unet_eval", keys=np.array(keys), shapes=np.array([str(s) for s in shapes]),
         n_params=sum(p.numel() for p in net.parameters()), **out)


def gen_convert():
    import utils.convert as ref_conv

    H, W, K = 64, 1024, (2.0, 26.9)
    depth, inten, cloud = convert_inputs(H, W)
    pts = ref_conv.pano_to_lidar_with_intensities(depth, inten, K)
    pts3 = ref_conv.pano_to_lidar(depth, K)
    Hs, Ws = 32, 256
    pano, pint = ref_conv.lidar_to_pano_with_intensities(cloud, Hs, Ws, K, max_depth=80)
    # round trip of the reference at full size: points of a range image fall back into their own pixels
    pano_rt, pint_rt = ref_conv.lidar_to_pano_with_intensities(pts.astype(np.float32)[::16], H, W, K)
    assert pts.dtype == np.float32 and np.array_equal(pts[:, :3], pts3)
    # inputs are regenerated from the same deterministic formulas by the tests (oracle.detparams.convert_inputs): only outputs are stored
    save("convert", H=H, W=W, K=np.array(K), pts=pts, Hs=Hs, Ws=Ws, pano=pano.astype(np.float32), pint=pint.astype(np.float32),
         pano_rt=pano_rt.astype(np.float32), pint_rt=pint_rt.astype(np.float32))


def gen_kitti360():
    """The reference's KITTI360Dataset on a tiny synthetic sequence written by oracle.detparams.write_kitti360_fixture."""
    import tempfile

    import data.kitti360_dataset as ref_ds
    from oracle.detparams import write_kitti360_fixture

    root = tempfile.mkdtemp(prefix="l4d_k360_")
    cfg = write_kitti360_fixture(root)
    out = {}
    for split, n_rays in (("train", 48), ("val", 48)):
        ds = ref_ds.KITTI360Dataset(device="cpu", split=split, root_path=root, sequence_id=cfg["sequence_id"], preload=True,
                                    scale=cfg["scale"], offset=cfg["offset"], fp16=False, num_rays_lidar=n_rays,
                                    fov_lidar=cfg["fov_lidar"])
        torch.manual_seed(11)
        b = ds.collate([1])
        out.update({f"{split}_poses": ds.poses_lidar, f"{split}_images": ds.images_lidar, f"{split}_times": ds.times,
                    f"{split}_rays_o": b["rays_o_lidar"], f"{split}_rays_d": b["rays_d_lidar"],
                    f"{split}_batch_images": b["images_lidar"], f"{split}_time": b["time"],
                    f"{split}_len": len(ds), f"{split}_num_rays": ds.num_rays_lidar})
    half = ref_ds.KITTI360Dataset(device="cpu", split="train", root_path=root, sequence_id=cfg["sequence_id"], preload=True,
                                  scale=cfg["scale"], offset=cfg["offset"], fp16=True, num_rays_lidar=16,
                                  fov_lidar=cfg["fov_lidar"])
    out["half_images"] = half.images_lidar.float()
    save("kitti360_reader", **out)
    shutil.rmtree(root, ignore_errors=True)


def gen_random_rays():
    """get_lidar_rays with N > 0: the reference's random pixel draws (single pixels, patches, anywhere) for fixed seeds."""
    import data.base_dataset as ref_data

    c, s = np.cos(0.4), np.sin(0.4)
    pose = torch.tensor([[[c, -s, 0, 0.3], [s, c, 0, -0.1], [0, 0, 1, 0.05], [0, 0, 0, 1]]], dtype=torch.float32)
    out = {"pose": pose}
    for tag, (H, W, N, patch) in {"p1": (16, 64, 96, 1), "p2": (16, 64, 96, 2), "p24": (16, 64, 96, [2, 4]), "any": (16, 64, 50, 0),
                                  "wrap": (8, 8, 48, [2, 4])}.items():
        torch.manual_seed(21)
        r = ref_data.get_lidar_rays(pose, [2.0, 26.9], H, W, N, patch)
        out[f"inds_{tag}"], out[f"rays_d_{tag}"] = r["inds"], r["rays_d"]
        out[f"cfg_{tag}"] = np.array([H, W, N] + (patch if isinstance(patch, list) else [patch, patch]))
    save("rays_random", **out)


def gen_param_order():
    """Registration order of the reference model's parameters and state-dict keys: positional formats (torch.optim state,
    torch_ema shadow lists) in its checkpoints rely on it."""
    import model.lidar4d as ref_lidar4d
    from oracle.make_golden import SMALL_MODEL

    ref = ref_lidar4d.LiDAR4D(**SMALL_MODEL)
    save("param_order", names=np.array([n for n, _ in ref.named_parameters()]), state_keys=np.array(list(ref.state_dict().keys())))


C2_LIKE = dict(n_levels_hash=16, n_levels_plane=4, min_resolution=8, num_layers_sigma=3, density_scale=30.0)


def gen_c2_like():
    """BASELINE configs[1] shape (L = 16 hash levels, 3-layer sigma network) through the reference's glue: density,
    flow and a render with gradient digests, so the oracle is pinned for this configuration as well."""
    import model.lidar4d as ref_lidar4d
    from oracle import tcnn_ref
    from oracle.detparams import fill_model, grad_digest
    from oracle.make_golden import SMALL_MODEL, test_rays

    tcnn_ref.set_precision("fp32")
    cfg = dict(SMALL_MODEL, **C2_LIKE)
    model = fill_model(ref_lidar4d.LiDAR4D(**cfg), seed=9)
    pts = det_uniform((256, 3), "c2pts", -1.0, 1.0)
    arrays = {}
    for fi in (0, 30):
        model.zero_grad()
        t = torch.tensor([[fi / 50]])
        out = model.density(pts, t)
        gsig, ggeo = det_uniform((256,), f"c2gs{fi}", -1, 1), det_uniform((256, 15), f"c2gg{fi}", -1, 1)
        ((out["sigma"] * gsig).sum() + (out["geo_feat"] * ggeo).sum()).backward()
        arrays[f"sigma_f{fi}"], arrays[f"geo_f{fi}"] = out["sigma"], out["geo_feat"]
        for n, v in grad_digest(model).items():
            arrays[f"gdig_f{fi}.{n}"] = v
    ro, rd = test_rays(24, 8)
    noise = det_uniform((24, 96), "c2n", 0.0, 1.0)
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: noise.clone()
    try:
        model.zero_grad()
        out = model.render(ro, rd, torch.tensor([[0.6]]), staged=False, num_steps=96, perturb=True)
    finally:
        torch.rand = orig_rand
    gd_, gi_ = det_uniform(tuple(out["depth_lidar"].shape), "c2gd", -1, 1), det_uniform(tuple(out["image_lidar"].shape), "c2gi", -1, 1)
    ((out["depth_lidar"] * gd_).sum() + (out["image_lidar"] * gi_).sum()).backward()
    for n, v in grad_digest(model).items():
        arrays[f"gdig_render.{n}"] = v
    save("c2_like", pts=pts, rays_o=ro, rays_d=rd, noise=noise, depth=out["depth_lidar"], image=out["image_lidar"],
         weights=out["weights"], z_vals=out["z_vals"], **arrays)


VARIANTS = {"frames4": (dict(num_frames=4), 2 / 3), "active": (dict(active_sensor=True), 0.5),
            "tres4": (dict(time_resolution=4, num_frames=9), 0.375)}


def gen_variants():
    """Configurations off the default path (few frames, active sensor, another time resolution) through the reference's
    glue: the render of tests/test_gpu_model.py::test_render_variants_vs_oracle, so the oracle is pinned there too."""
    import model.lidar4d as ref_lidar4d
    from oracle import tcnn_ref
    from oracle.detparams import fill_model, grad_digest
    from oracle.make_golden import SMALL_MODEL, test_rays

    tcnn_ref.set_precision("fp32")
    ro, rd = test_rays(16, 3)   # small: the CPU suite re-renders this with the oracle
    noise = det_uniform((16, 64), "vn", 0.0, 1.0)
    arrays = {"rays_o": ro, "rays_d": rd, "noise": noise}
    for tag, (kw, frame_t) in VARIANTS.items():
        model = fill_model(ref_lidar4d.LiDAR4D(**dict(SMALL_MODEL, density_scale=30.0, **kw)), seed=5)
        orig_rand = torch.rand
        torch.rand = lambda *a, **k: noise.clone()
        try:
            out = model.render(ro, rd, torch.tensor([[frame_t]], dtype=torch.float32), staged=False, num_steps=64, perturb=True)
        finally:
            torch.rand = orig_rand
        gd_, gi_ = det_uniform((1, 16), "vgd", -1, 1), det_uniform((1, 16, 2), "vgi", -1, 1)
        ((out["depth_lidar"] * gd_).sum() + (out["image_lidar"] * gi_).sum()).backward()
        arrays.update({f"{tag}.depth": out["depth_lidar"], f"{tag}.image": out["image_lidar"], f"{tag}.wsum": out["weights_sum_lidar"],
                       f"{tag}.z_vals_sum": out["z_vals"].double().sum()})
        for n, v in grad_digest(model).items():
            arrays[f"{tag}.gdig.{n}"] = v
    save("render_variants", **arrays)


def gen_preprocess(scratch):
    """The reference's preprocessing functions (raw scans -> range views; range views + poses -> scene scale / offset) on
    the tiny deterministic sequence of oracle.detparams.write_scan_fixture."""
    import os
    import sys
    import tempfile

    from oracle.detparams import write_scan_fixture
    sys.path.insert(0, os.path.join(scratch, "ref", "data", "preprocess"))
    import cal_seq_config as ref_cfg
    import generate_rangeview as ref_gen

    H, W, K = 16, 64, (2.0, 26.9)
    root = tempfile.mkdtemp(prefix="l4d_pre_")
    bins, poses = write_scan_fixture(root)
    views, view_paths = [], []
    for p in bins:
        cloud = np.fromfile(p, dtype=np.float32).reshape(-1, 4)
        view = ref_gen.LiDAR_2_Pano_KITTI(cloud, H, W, K)
        views.append(view)
        view_paths.append(p.replace(".bin", ".npy"))
        np.save(view_paths[-1], view)
    scale, center = ref_cfg.cal_centerpose_bound_scale(view_paths, poses, list(K))
    save("preprocess", H=H, W=W, K=np.array(K), views=np.stack(views), poses=np.stack(poses), scale=scale, centerpose=np.array(center))
    shutil.rmtree(root, ignore_errors=True)


def main():
    torch.set_num_threads(8)
    R = _import_reference()
    gen_unet()
    gen_convert()
    gen_kitti360()
    gen_random_rays()
    gen_param_order()
    gen_c2_like()
    gen_variants()
    gen_preprocess(R["scratch"])
    shutil.rmtree(R["scratch"], ignore_errors=True)


if __name__ == "__main__":
    main()
