"""SURVEY 8f "next" rows 3 and 4: U-Net ray-drop refinement and the range-image <-> point-cloud conversions, against
fixtures produced by the reference's own code (oracle/make_golden_next.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import convert_ref
from oracle.detparams import convert_inputs, fill_unet

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _unet():
    from lidar4d_amd.unet import UNet
    return fill_unet(UNet(in_channels=3, out_channels=1), seed=0).eval()


def _check_unet(dev, tol):
    g = np.load(os.path.join(GOLD, "unet_eval.npz"))
    net = _unet().to(dev)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys"]), "state-dict keys (and their order) must equal the reference's"
    assert [str(list(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"])
    for tag in ("a", "b"):
        x = torch.from_numpy(g["x_" + tag]).to(dev).requires_grad_(True)
        y = net(x)
        assert y.shape == g["y_" + tag].shape
        np.testing.assert_allclose(y.detach().cpu().numpy(), g["y_" + tag], atol=tol, rtol=0)
        net.zero_grad()
        (y * torch.from_numpy(g["gy_" + tag]).to(dev)).sum().backward()
        scale = np.abs(g["gx_" + tag]).max()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g["gx_" + tag], atol=10 * tol * scale, rtol=0)
        gw = g["gw_inc_" + tag]
        np.testing.assert_allclose(net.inc.conv.weight.grad.cpu().numpy(), gw, atol=10 * tol * np.abs(gw).max(), rtol=0)
        ga = net.attn.proj_qkv.weight.grad.double()
        np.testing.assert_allclose([ga.sum().item(), ga.abs().sum().item()], g["gw_attn_digest_" + tag],
                                   rtol=100 * tol, atol=1e-6)


def test_unet_matches_reference_cpu():
    _check_unet("cpu", 2e-6)


@pytest.mark.gpu
def test_unet_matches_reference_gpu():
    # MIOpen convolutions (fp32): different summation order than the CPU run that made the fixture
    _check_unet("cuda", 2e-5)


@pytest.mark.gpu
def test_model_carries_unet_and_refines():
    from lidar4d_amd import LiDAR4D
    from oracle.make_golden import SMALL_MODEL
    m = LiDAR4D(**SMALL_MODEL).cuda()
    assert any(k.startswith("unet.inc.conv") for k in m.state_dict())
    # runner.py:413-416: refine the stacked [raydrop, intensity, depth] image
    img = torch.rand(3, 32, 64, device="cuda")
    m.unet.eval()
    out = m.unet(img.unsqueeze(0)).squeeze(0)
    assert out.shape == (1, 32, 64) and float(out.detach().min()) >= 0 and float(out.detach().max()) <= 1
    # the render optimiser groups do not contain the U-Net (lidar4d.py:226-237; it has its own Adam, runner.py:872)
    ids = {id(p) for g in m.get_params(1e-2) for p in g["params"]}
    assert not any(id(p) in ids for p in m.unet.parameters())


def test_convert_oracle_pinned_to_reference():
    g = np.load(os.path.join(GOLD, "convert.npz"))
    depth, inten, cloud = convert_inputs(int(g["H"]), int(g["W"]))
    K = tuple(g["K"])
    pts = convert_ref.pano_to_lidar_with_intensities(depth, inten, K)
    assert pts.dtype == np.float32 and np.array_equal(pts, g["pts"])
    pano, pint = convert_ref.lidar_to_pano_with_intensities(cloud, int(g["Hs"]), int(g["Ws"]), K)
    assert np.array_equal(pano, g["pano"]) and np.array_equal(pint, g["pint"])
    pano, pint = convert_ref.lidar_to_pano_with_intensities(g["pts"][::16], int(g["H"]), int(g["W"]), K)
    assert np.array_equal(pano, g["pano_rt"]) and np.array_equal(pint, g["pint_rt"])


@pytest.mark.gpu
def test_pano_to_lidar_gpu():
    from lidar4d_amd import convert
    g = np.load(os.path.join(GOLD, "convert.npz"))
    depth, inten, _ = convert_inputs(int(g["H"]), int(g["W"]))
    K = tuple(g["K"])
    pts = convert.pano_to_lidar_with_intensities(torch.from_numpy(depth).cuda(), torch.from_numpy(inten).cuda(), K)
    ref = g["pts"]
    assert pts.shape == ref.shape, "same pixels kept, same (row-major) order"
    out = pts.cpu().numpy()
    assert np.array_equal(out[:, 3], ref[:, 3]), "intensities are copied: bit-exact, which also proves the order"
    # coordinates: fp32 sin/cos of the device library vs numpy's, <= 2 ulp of the range
    np.testing.assert_allclose(out[:, :3], ref[:, :3], rtol=0, atol=3e-7 * 80)
    p3 = convert.pano_to_lidar(torch.from_numpy(depth).cuda(), K)
    assert torch.equal(p3, pts[:, :3])
    buf, count = convert.pano_to_lidar_with_intensities(torch.from_numpy(depth).cuda(), None, K, return_count=True)
    assert int(count) == ref.shape[0] and buf.shape == (depth.size, 4) and float(buf[: int(count), 3].abs().max()) == 0
    # empty and full images
    z = torch.zeros(8, 16, device="cuda")
    assert convert.pano_to_lidar(z, K).shape == (0, 3)
    assert convert.pano_to_lidar(z + 1, K).shape == (128, 3)


@pytest.mark.gpu
def test_lidar_to_pano_gpu():
    from lidar4d_amd import convert
    g = np.load(os.path.join(GOLD, "convert.npz"))
    _, _, cloud = convert_inputs(int(g["H"]), int(g["W"]))
    K = tuple(g["K"])
    for pts, H, W, pano_ref, pint_ref in ((cloud, int(g["Hs"]), int(g["Ws"]), g["pano"], g["pint"]),
                                          (g["pts"][::16], int(g["H"]), int(g["W"]), g["pano_rt"], g["pint_rt"])):
        pano, pint = convert.lidar_to_pano_with_intensities(torch.from_numpy(np.ascontiguousarray(pts)).cuda(), H, W, K)
        pano, pint = pano.cpu().numpy(), pint.cpu().numpy()
        # a point whose angle sits within an ulp of a pixel boundary may land in the neighbouring pixel (device atan2f
        # vs libm's); everything else must be bit-identical
        bad = (pano != pano_ref) | (pint != pint_ref)
        assert bad.mean() < 2e-3, f"{bad.sum()} of {bad.size} pixels differ"
    # empty cloud -> empty image
    pano, pint = convert.lidar_to_pano_with_intensities(torch.zeros(0, 4, device="cuda"), 8, 16, K)
    assert float(pano.abs().max()) == 0 and float(pint.abs().max()) == 0
    # pano -> points -> pano is the identity on the occupied pixels (size-independent property, full 64 x 2048 frame)
    H, W = 64, 2048
    depth = torch.rand(H, W, device="cuda") * 70 + 2
    depth[torch.rand(H, W, device="cuda") < 0.1] = 0
    inten = torch.rand(H, W, device="cuda")
    pts = convert.pano_to_lidar_with_intensities(depth, inten, K)
    back, iback = convert.lidar_to_pano_with_intensities(pts, H, W, K)
    occupied = depth != 0
    same = (back - depth).abs() <= 1e-5 * depth
    # (column 0 sits on the atan2 branch cut, beta = +-pi: the reference's projection sends half of it to c = W, out of bounds)
    assert float(same[:, 1:][occupied[:, 1:]].float().mean()) > 0.999
    assert float((iback == inten)[:, 1:][occupied[:, 1:]].float().mean()) > 0.999


@pytest.mark.gpu
def test_points_meter_and_test_step_gpu():
    """utils/metrics.py:249-270 + runner.py:438-470 on the device, against the CPU restatements."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_FOV, KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.metrics import PointsMeter
    from lidar4d_amd.trainer import Trainer
    from oracle import chamfer_ref
    from oracle.make_golden import SMALL_MODEL
    torch.manual_seed(0)
    H, W = 16, 128
    data = SyntheticKitti360("cuda", H=H, W=W, num_frames=5, num_rays=256)
    model = LiDAR4D(**dict(SMALL_MODEL, num_frames=5, near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE)).cuda().eval()
    tr = Trainer(model, data, num_steps=96)
    fr = data.frame(2)
    assert fr["rays_d_lidar"].shape == (1, H * W, 3) and fr["images_lidar"].shape == (1, H, W, 3)
    rd, ri, dep = tr.test_step(fr, refine=True, max_ray_batch=300)   # ragged last chunk
    assert rd.shape == (1, H, W) and ri.shape == (1, H, W) and dep.shape == (1, H, W)
    # staged == one un-chunked call (rays are independent), and the masking rule
    with torch.no_grad():
        full = model.render(fr["rays_o_lidar"], fr["rays_d_lidar"], fr["time"], staged=False, perturb=False, num_steps=96)
        img = full["image_lidar"].reshape(1, H, W, 2)
        stacked = torch.cat([img[..., 0], img[..., 1], full["depth_lidar"].reshape(1, H, W)], 0).unsqueeze(0)
        rd_ref = model.unet(stacked).squeeze(0)
    assert torch.allclose(rd, rd_ref, atol=1e-5)
    keep = (rd_ref > 0.5).float()
    assert torch.allclose(dep, full["depth_lidar"].reshape(1, H, W) * keep, atol=1e-6)
    rd2, _, dep2 = tr.test_step(fr, refine=False)
    assert torch.allclose(rd2, img[..., 0], atol=1e-6)
    # PointsMeter vs numpy conversion + brute-force chamfer
    gt = fr["images_lidar"]
    gt_depth = gt[..., 2] * gt[..., 0]
    pred = gt_depth * (1 + 0.05 * torch.rand_like(gt_depth))
    pred[0, :2] = 0  # some dropped rows
    meter = PointsMeter(scale=KITTI360_SCALE, intrinsics=KITTI360_FOV)
    meter.update(pred, gt_depth)
    meter.update(gt_depth, gt_depth)
    p = convert_ref.pano_to_lidar_with_intensities((pred[0] / KITTI360_SCALE).cpu().numpy(), np.zeros((H, W), np.float32), KITTI360_FOV)[:, :3]
    q = convert_ref.pano_to_lidar_with_intensities((gt_depth[0] / KITTI360_SCALE).cpu().numpy(), np.zeros((H, W), np.float32), KITTI360_FOV)[:, :3]
    d1, d2, _, _ = chamfer_ref.chamfer(torch.from_numpy(p)[None], torch.from_numpy(q)[None])
    cd = float(d1.mean() + d2.mean())
    p1, p2 = float((d1 < 0.05).float().mean()), float((d2 < 0.05).float().mean())
    f = 2 * p1 * p2 / (p1 + p2)
    v = torch.stack(meter.V).cpu().numpy()
    np.testing.assert_allclose(v[0], [cd, f], rtol=2e-4)
    np.testing.assert_allclose(v[1], [0.0, 1.0], atol=1e-9)
    assert meter.measure().shape == (2,)


@pytest.mark.gpu
def test_flow_loss_vs_oracle():
    """runner.py:222-253 (scene-flow consistency) through model.flow + the HIP chamfer, against the CPU restatement."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import Trainer, flow_loss, process_pointcloud
    from oracle import chamfer_ref, fields_ref, tcnn_ref
    from oracle.detparams import fill_model, grad_digest
    from oracle.make_golden import SMALL_MODEL
    prev = tcnn_ref.get_precision()
    tcnn_ref.set_precision("tcnn")
    try:
        cfg = dict(SMALL_MODEL, num_frames=5, near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE)
        ref = fill_model(fields_ref.LiDAR4D(**cfg), seed=3, flow_out_amp=0.002)
        hip = fill_model(LiDAR4D(**cfg), seed=3, flow_out_amp=0.002).cuda()
        data = SyntheticKitti360("cuda", H=16, W=64, num_frames=5, num_rays=128)
        pcs, grounds = process_pointcloud(data)
        assert set(pcs) == {str(k) for k in range(5)} and all(v.shape[1] == 3 for v in pcs.values())
        n_pts = sum(v.shape[0] + grounds[k].shape[0] for k, v in pcs.items())
        assert n_pts == int((data.images[..., 0] > 0).sum()), "every returned pixel is either ground or non-ground"
        t = torch.tensor([[2 / 4]])
        tg = torch.tensor([0.37])
        hip.zero_grad()
        loss = flow_loss(hip, pcs, grounds, t.cuda(), 5, t_ground=tg.cuda())
        loss.backward()
        # oracle: same formula on the CPU restatement
        ref.zero_grad()
        pc = pcs["2"].cpu()
        pred = ref.flow(pc, t)
        loss_ref = pc.new_zeros(())
        for step in (1, 2):
            for sign, key in ((+1, "forward"), (-1, "backward")):
                other = pcs.get(str(2 + sign * step))
                if other is None:
                    continue
                d1, d2, _, _ = chamfer_ref.chamfer((pc + pred[key] * step)[None], other.cpu()[None])
                loss_ref = loss_ref + (d1.sum() + d2.sum()) * 0.5
        zf = ref.flow(grounds["2"].cpu(), tg.reshape(1, 1))
        loss_ref = loss_ref + 0.001 * (zf["forward"].abs().sum() + zf["backward"].abs().sum())
        loss_ref.backward()
        assert abs(float(loss) - float(loss_ref)) <= 2e-3 * abs(float(loss_ref)), (float(loss), float(loss_ref))
        dig_ref, dig = grad_digest(ref), grad_digest(hip)
        for n, v in dig_ref.items():
            if not n.startswith("flow_net."):
                assert abs(dig[n]).max() == 0
                continue
            scale = max(abs(v[1]), 1e-12)
            assert max(abs(dig[n][k] - v[k]) for k in range(3)) / scale < 3e-2, (n, dig[n], v)
        # a training step with all optional terms runs and changes the flow parameters
        tr = Trainer(hip, data, num_steps=64, chamfer=True, flow=True, init_scale=1.0)
        before = hip.flow_net.grid_enc.params.detach().clone()
        l0 = float(tr.train_step(data.batch_for(2)))
        assert np.isfinite(l0) and not torch.equal(before, hip.flow_net.grid_enc.params.detach())
    finally:
        tcnn_ref.set_precision(prev)


def test_kitti360_reader_matches_reference(tmp_path):
    """lidar4d_amd.kitti360.KITTI360Dataset on the reference's on-disk layout, against what the reference's own reader and
    collate produce for the same files and the same torch seed (oracle/make_golden_next.py::gen_kitti360).  Host logic:
    runs wherever torch runs."""
    from lidar4d_amd.kitti360 import KITTI360Dataset
    from oracle.detparams import write_kitti360_fixture
    g = np.load(os.path.join(GOLD, "kitti360_reader.npz"))
    cfg = write_kitti360_fixture(str(tmp_path))
    for split in ("train", "val"):
        ds = KITTI360Dataset(device="cpu", split=split, root_path=str(tmp_path), sequence_id=cfg["sequence_id"], preload=True,
                             scale=cfg["scale"], offset=cfg["offset"], fp16=False, num_rays_lidar=48, fov_lidar=cfg["fov_lidar"])
        assert len(ds) == int(g[f"{split}_len"]) and ds.num_rays_lidar == int(g[f"{split}_num_rays"])
        assert np.array_equal(ds.poses_lidar.numpy(), g[f"{split}_poses"]), "normalised sensor poses (frames sorted by file)"
        assert np.array_equal(ds.images_lidar.numpy(), g[f"{split}_images"]), "[ray-drop mask, intensity, depth * scale]"
        assert np.array_equal(ds.times.numpy(), g[f"{split}_times"])
        torch.manual_seed(11)
        b = ds.collate([1])
        assert b["H_lidar"] == 8 and b["W_lidar"] == 32 and b["poses_lidar"].shape == (1, 4, 4)
        # same seed -> same random pixels as the reference's collate (row in [0, H-1), column in [0, W))
        assert np.array_equal(b["images_lidar"].numpy(), g[f"{split}_batch_images"])
        assert np.array_equal(b["time"].numpy(), g[f"{split}_time"])
        np.testing.assert_allclose(b["rays_o_lidar"].numpy(), g[f"{split}_rays_o"], rtol=0, atol=0)
        np.testing.assert_allclose(b["rays_d_lidar"].numpy(), g[f"{split}_rays_d"], rtol=0, atol=1e-7)
    half = KITTI360Dataset(device="cpu", split="refine", root_path=str(tmp_path), sequence_id=cfg["sequence_id"], preload=True,
                           scale=cfg["scale"], offset=cfg["offset"], fp16=True, num_rays_lidar=16, fov_lidar=cfg["fov_lidar"])
    assert half.images_lidar.dtype == torch.half and np.array_equal(half.images_lidar.float().numpy(), g["half_images"])
    assert half.num_rays_lidar == -1 and not half.training and half.split == "train"  # 'refine': whole training frames
    loader = ds.dataloader()
    assert loader._data is ds and loader.has_gt and len(list(loader)) == len(ds)
    with pytest.raises(ValueError):
        KITTI360Dataset(root_path=str(tmp_path), sequence_id="42")


def test_random_ray_draws_match_reference():
    """get_lidar_rays(N > 0): same seed -> the reference's pixels, for single pixels, square and rectangular patches,
    unconstrained draws and patches that wrap around the panorama (base_dataset.py:36-70)."""
    from lidar4d_amd.data import get_lidar_rays
    g = np.load(os.path.join(GOLD, "rays_random.npz"))
    pose = torch.from_numpy(g["pose"])
    for tag in ("p1", "p2", "p24", "any", "wrap"):
        H, W, N, px, py = (int(v) for v in g[f"cfg_{tag}"])
        torch.manual_seed(21)
        r = get_lidar_rays(pose, [2.0, 26.9], H, W, N, px if px == py else [px, py])
        assert np.array_equal(r["inds"].numpy(), g[f"inds_{tag}"]), tag
        np.testing.assert_allclose(r["rays_d"].numpy(), g[f"rays_d_{tag}"], rtol=0, atol=1e-7)
    with pytest.raises(ValueError):
        get_lidar_rays(pose, [2.0, 26.9], 16, 64, 50, 4)  # 50 rays cannot be cut into 4 x 4 patches


@pytest.mark.gpu
def test_training_step_with_all_loss_terms():
    """Trainer with the ray-chamfer, scene-flow and line-of-sight terms: the extra terms reach the field through
    d(depth) and d(weights) of the fused backward; the step must run, stay finite and move the parameters."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import Trainer
    from oracle.detparams import fill_model
    from oracle.make_golden import SMALL_MODEL
    cfg = dict(SMALL_MODEL, num_frames=5, near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE, density_scale=20.0)
    data = SyntheticKitti360("cuda", H=16, W=64, num_frames=5, num_rays=128)
    losses = {}
    off = dict(chamfer=False, flow=False)  # the Trainer's defaults are the reference's: chamfer and scene-flow terms on
    for name, kw in (("plain", off), ("urf", dict(off, urf=True)), ("all", dict(urf=True))):
        m = fill_model(LiDAR4D(**cfg), seed=3, flow_out_amp=0.002).cuda()
        tr = Trainer(m, data, num_steps=64, iters=10, init_scale=1.0, **kw)  # scale 1: no skipped first steps
        before = m._store.flat.clone()
        data.gen.manual_seed(0)
        torch.manual_seed(0)
        losses[name] = [float(tr.train_step(data.batch_for(2))) for _ in range(2)]
        assert all(np.isfinite(losses[name])) and not torch.equal(before, m._store.flat)
    assert losses["urf"][0] > losses["plain"][0] and losses["all"][0] > losses["urf"][0]  # the terms are non-negative
    # rays drawn as 2 x 8 pixel patches switch the depth-gradient term on (runner.py:277-367, 700-705)
    m = fill_model(LiDAR4D(**cfg), seed=3, flow_out_amp=0.002).cuda()
    tr = Trainer(m, data, num_steps=64, iters=10, init_scale=1.0)
    data.patch_size_lidar = [2, 8]
    try:
        b = data.batch_for(2)
        inds = b["rays_d_lidar"].shape[1]
        assert inds == 128
        l_patch = float(tr.train_step(b))
    finally:
        data.patch_size_lidar = 1
    assert np.isfinite(l_patch)


def test_preprocess_glue_matches_reference_scripts(tmp_path):
    """lidar4d_amd.preprocess (raw scans -> range views; range views + poses -> scene scale / offset / config file) against
    the reference's data/preprocess functions run on the same files (gen_preprocess).  The conversions are injected from
    the CPU restatement here -- the HIP versions have their own GPU parity tests -- so this pins the file handling and
    the float64 reductions."""
    from lidar4d_amd import preprocess
    from oracle.detparams import write_scan_fixture
    g = np.load(os.path.join(GOLD, "preprocess.npz"))
    H, W, K = int(g["H"]), int(g["W"]), tuple(float(v) for v in g["K"])
    bins, poses = write_scan_fixture(str(tmp_path))
    assert np.array_equal(np.stack(poses), g["poses"])
    to_pano = lambda pts, h, w, k, md: convert_ref.lidar_to_pano_with_intensities(pts.numpy(), h, w, k, md)
    written = preprocess.generate_rangeview(bins, str(tmp_path / "train"), H, W, K, device="cpu", to_pano=to_pano)
    assert [os.path.basename(p) for p in written] == [f"{k:010d}.npy" for k in range(3)]
    for k, p in enumerate(written):
        view = np.load(p)
        assert view.dtype == np.float64 and view.shape == (H, W, 3) and np.array_equal(view, g["views"][k])
    to_points = lambda depth, fov: convert_ref.pano_to_lidar_with_intensities(depth.numpy(), np.zeros_like(depth.numpy()), fov)[:, :3]
    scale, center, near, far = preprocess.cal_centerpose_bound_scale(written, poses, list(K), device="cpu", to_points=to_points)
    assert abs(scale - float(g["scale"])) <= 1e-12 * float(g["scale"])
    np.testing.assert_allclose(center, g["centerpose"], rtol=1e-12)
    assert 0 < near < far < 80.0
    cfg = preprocess.write_seq_config(str(tmp_path / "configs" / "kitti360_4950.txt"), "kitti360", "data/kitti360", "4950", 51, K, scale, center)
    lines = open(cfg).read().splitlines()
    assert lines[0] == "dataloader = kitti360" and lines[3] == "num_frames = 51" and lines[5] == f"scale = {scale}"
    assert lines[4] == "fov_lidar = [2.0, 26.9]" and lines[6].startswith("offset = [")


def test_transforms_writer_roundtrips_through_the_reader(tmp_path):
    """preprocess.load_lidar_poses + write_transforms (kitti360_loader.py:62-127, kitti360_to_nerf.py:76-146) produce files
    the (reference-pinned) reader loads: schema, splits, pose algebra."""
    import json
    from lidar4d_amd import preprocess
    from lidar4d_amd.kitti360 import KITTI360Dataset, SEQUENCE_FRAMES
    root, seq = tmp_path / "kitti360", "4950"
    first, last = SEQUENCE_FRAMES[seq]
    k3 = root / "KITTI-360"
    (k3 / "calibration").mkdir(parents=True)
    (k3 / "data_poses" / "2013_05_28_drive_0000_sync").mkdir(parents=True)
    cam_to_imu = np.array([[0.0, 0.0, 1.0, 1.5], [-1.0, 0.0, 0.0, 0.1], [0.0, -1.0, 0.0, 0.9]])
    cam_to_velo = np.array([[0.0, -1.0, 0.0, 0.2], [0.0, 0.0, -1.0, -0.1], [1.0, 0.0, 0.0, -0.3]])
    (k3 / "calibration" / "calib_cam_to_pose.txt").write_text(
        "image_00: " + " ".join(f"{v:.6f}" for v in cam_to_imu.reshape(-1)) + "\nimage_01: " + " ".join(["0"] * 12) + "\n")
    (k3 / "calibration" / "calib_cam_to_velo.txt").write_text(" ".join(f"{v:.6f}" for v in cam_to_velo.reshape(-1)) + "\n")
    rows = []
    for fid in range(first, last + 1):
        if fid == first + 7:
            continue  # a frame without a pose reuses the previous one
        a = 0.01 * (fid - first)
        imu = np.array([[np.cos(a), -np.sin(a), 0.0, 1000.0 + (fid - first)], [np.sin(a), np.cos(a), 0.0, 3700.0], [0.0, 0.0, 1.0, 115.0]])
        rows.append([fid] + imu.reshape(-1).tolist())
    np.savetxt(k3 / "data_poses" / "2013_05_28_drive_0000_sync" / "poses.txt", np.array(rows))
    poses = preprocess.load_lidar_poses(str(k3), "2013_05_28_drive_0000", range(first, last + 1))
    assert poses.shape == (51, 4, 4) and np.array_equal(poses[7], poses[6])
    pad = lambda m: np.vstack([m, [0, 0, 0, 1]])
    want = pad(np.array(rows[3][1:]).reshape(3, 4)) @ pad(cam_to_imu) @ np.linalg.inv(pad(cam_to_velo))
    np.testing.assert_allclose(poses[3], want, atol=1e-9)
    (root / "train").mkdir()
    H, W = 4, 16
    for fid in range(first, last + 1):
        view = np.zeros((H, W, 3))
        view[:, :, 1], view[:, :, 2] = 0.5, 10.0 + (fid - first)
        np.save(root / "train" / f"{fid:010d}.npy", view)
    paths = preprocess.write_transforms(str(root), seq, poses)
    assert [os.path.basename(p) for p in paths] == [f"transforms_4950_{s}.json" for s in ("train", "val", "test")]
    doc = json.load(open(paths[1]))
    assert doc["num_frames"] == 51 and doc["num_frames_split"] == 4 and [f["frame_id"] for f in doc["frames"]] == [4960, 4970, 4980, 4990]
    assert json.load(open(paths[0]))["num_frames_split"] == 47 and doc["h_lidar"] == H and doc["w_lidar"] == W
    ds = KITTI360Dataset(device="cpu", split="val", root_path=str(root), sequence_id=seq, preload=False, scale=0.01,
                         offset=[1000.0, 3700.0, 115.0], fp16=False, fov_lidar=[2.0, 26.9])
    assert len(ds) == 4 and ds.H_lidar == H and ds.W_lidar == W
    np.testing.assert_allclose(ds.times.numpy().reshape(-1), [0.2, 0.4, 0.6, 0.8], rtol=1e-6)
    np.testing.assert_allclose(ds.images_lidar[0, 0, 0].numpy(), [1.0, 0.5, 20.0 * 0.01], rtol=1e-6)
    np.testing.assert_allclose(ds.poses_lidar[0, :3, 3].numpy(), (poses[10, :3, 3] - np.array([1000.0, 3700.0, 115.0])) * 0.01, atol=1e-6)
