"""The fused step glue (csrc/glue.hip: batch assembly, primary losses + ray chamfer, scene-flow loss) against the torch
restatements of the reference's formulas they replace (``-m gpu``, MI355X).  The restatements themselves are what
tests/test_next_rows.py / test_oracle_golden.py pin against the reference (get_lidar_rays goldens, flow loss vs oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_fused_ray_batch_equals_get_lidar_rays():
    """SyntheticKitti360.batch_for: one launch behind the two random draws == data.get_lidar_rays + gather (base_dataset.py:36-102,
    kitti360_dataset.py:181-187) for the same generator state: same pixels, same ground truth, directions to 1 ulp-ish (the
    torch path multiplies by the rotation with a batched GEMM)."""
    from lidar4d_amd.data import SyntheticKitti360
    a = SyntheticKitti360(DEV, H=64, W=1024, num_frames=5, num_rays=4096, seed=9)
    b = SyntheticKitti360(DEV, H=64, W=1024, num_frames=5, num_rays=4096, seed=9)
    b.fused_batch = False
    # a rotated, translated pose (the synthetic track itself is axis-aligned)
    c, s = np.cos(0.3), np.sin(0.3)
    pose = torch.tensor([[c, -s, 0, 0.1], [s, c, 0, -0.2], [0, 0, 1, 0.03], [0, 0, 0, 1]], dtype=torch.float32, device=DEV)
    a.poses[2], b.poses[2] = pose, pose
    for frame in (2, 0, 2):
        x, y = a.batch_for(frame), b.batch_for(frame)
        for k in ("rays_o_lidar", "rays_d_lidar", "images_lidar", "time"):
            assert x[k].shape == y[k].shape and x[k].dtype == y[k].dtype, k
        assert torch.equal(x["images_lidar"], y["images_lidar"])          # same pixels drawn, same gather
        assert torch.equal(x["rays_o_lidar"], y["rays_o_lidar"])
        assert float((x["rays_d_lidar"] - y["rays_d_lidar"]).abs().max()) <= 2e-7
        assert x["time_host"] == y["time_host"] and torch.equal(x["time"], y["time"])
    # the device generators stayed in step
    assert torch.equal(torch.randint(0, 1000, [8], device=DEV, generator=a.gen), torch.randint(0, 1000, [8], device=DEV, generator=b.gen))


@pytest.mark.parametrize("chamfer,world", [(False, 1), (True, 1), (True, 4)])
def test_fused_primary_losses_equal_torch_losses(chamfer, world):
    """trainer.primary_losses (l4d_lidar_losses [+ l4d_chamfer_fwd + l4d_ray_chamfer_grad], backward = l4d_scale_buffers) ==
    lidar_loss [+ ray_chamfer_loss / world] (runner.py:179-220) in value and in the gradients wrt the rendered depth / image,
    under an upstream gradient (the loss scale) that is not 1."""
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import lidar_loss, primary_losses, ray_chamfer_loss
    data = SyntheticKitti360(DEV, H=32, W=256, num_frames=3, num_rays=2048, seed=4)
    batch = data.batch_for(1)
    g = torch.Generator(device=DEV).manual_seed(1)
    gt = batch["images_lidar"]
    # a prediction near the ground truth with exact hits (|error| = 0 -> sign 0) and dropped rays in it
    depth0 = (gt[..., 2] + 0.02 * torch.randn(gt.shape[:2], device=DEV, generator=g)).clamp_min(0.0)
    depth0[:, :64] = gt[:, :64, 2]
    image0 = torch.rand(gt.shape[:2] + (2,), device=DEV, generator=g)
    res = {}
    for name in ("torch", "fused"):
        depth, image = depth0.clone().requires_grad_(True), image0.clone().requires_grad_(True)
        out = {"depth_lidar": depth, "image_lidar": image}
        if name == "torch":
            loss = lidar_loss(out, gt, scale=KITTI360_SCALE)
            if chamfer:
                loss = loss + ray_chamfer_loss(out, batch, KITTI360_SCALE) / world
        else:
            loss = primary_losses(out, batch, KITTI360_SCALE, chamfer=chamfer, world=world)
        (loss * 1024.0).backward()
        res[name] = (float(loss), depth.grad.clone(), image.grad.clone())
    lt, lf = res["torch"][0], res["fused"][0]
    assert abs(lt - lf) <= 1e-5 * abs(lt), (lt, lf)
    for k, what in ((1, "d/d depth"), (2, "d/d image")):
        a, b = res["fused"][k], res["torch"][k]
        assert a.shape == b.shape
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        assert err <= 2e-5, (what, err)
    assert float(res["fused"][1].abs().max()) > 0


def test_trainer_step_fused_equals_torch_losses():
    """One training step from the same state and batch with the fused glue and with the torch restatement of the losses:
    same loss, same parameters afterwards (to the rounding of the loss gradients)."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import SyntheticKitti360
    from lidar4d_amd.trainer import Trainer
    from oracle.detparams import fill_model
    from oracle.make_golden import SMALL_MODEL
    cfg = dict(SMALL_MODEL, density_scale=20.0, num_frames=5)
    outs = {}
    for fused in (True, False):
        data = SyntheticKitti360(DEV, H=16, W=64, num_frames=5, num_rays=256, seed=3)
        data.fused_batch = fused
        model = fill_model(LiDAR4D(**cfg), seed=11).to(DEV)
        tr = Trainer(model, data, num_steps=64, chamfer=True, flow=True, init_scale=1.0)
        tr.fused_losses = fused
        tr.fused_flow_loss = fused
        torch.manual_seed(5)  # (perturbation noise and the ground-point time of the scene-flow loss)
        loss = float(tr.train_step(data.batch_for(2)))
        outs[fused] = (loss, model._store.flat.detach().clone())
    (la, pa), (lb, pb) = outs[True], outs[False]
    assert np.isfinite(la) and abs(la - lb) <= 1e-4 * abs(lb), (la, lb)
    moved = (pb - fill_model(LiDAR4D(**cfg), seed=11).to(DEV)._store.flat).abs().max()
    assert float(moved) > 0
    # Adam's first step moves every touched parameter by ~lr whatever the gradient's size, so equal parameters = equal gradient signs
    # nearly everywhere; allow the few entries whose tiny gradients round differently
    differing = float(((pa - pb).abs() > 1e-4).float().mean())
    assert differing < 2e-3, differing


def test_flow_loss_on_its_own_stream_equals_single_stream(monkeypatch):
    """Trainer(flow_loss_stream=True): the scene-flow term's forward and backward run on a second stream next to the render path; its
    parameter gradients wait in private buffers and are added to the arena once the render path's own flow-field gradients are in
    (both write the same ranges with plain stores).  Same loss and -- compared BEFORE the optimiser touches them -- the same gradient
    arena as the one-stream step, several steps in a row from the same state (a race would show as a run-to-run difference), on a
    batch large enough that the two streams really overlap."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd import trainer as trainer_mod
    from lidar4d_amd.trainer import Trainer
    torch.manual_seed(0)
    model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(DEV)
    data = SyntheticKitti360(DEV, W=1024, num_rays=2048, seed=7, frame_seed=7)
    # (the ground points' random time is pinned: on its own stream the term is evaluated BEFORE the render draws its sample jitter,
    # i.e. the two orders consume the generator differently)
    tg, fl_fn = torch.tensor([0.37], device=DEV), trainer_mod.flow_loss
    monkeypatch.setattr(trainer_mod, "flow_loss", lambda *a, **kw: fl_fn(*a, **{**kw, "t_ground": tg}))
    batch = {k: (v.contiguous().clone() if torch.is_tensor(v) else v) for k, v in data.batch_for(20).items()}
    grads = {}
    for side in (False, True, True, False, True):
        tr = Trainer(model, data, iters=200, chamfer=True, flow=True, ema_decay=None, init_scale=1024.0, flow_loss_stream=side)
        got = []
        step = tr.opt.step
        tr.opt.step = lambda **kw: got.append(model._store.flat_grad.detach().clone())  # (no update: every run starts from the same state)
        torch.manual_seed(5)
        loss = float(tr.train_step(batch))
        tr.opt.step = step
        torch.cuda.synchronize()
        grads.setdefault(side, []).append((loss, got[0]))
    l0, g0 = grads[False][0]
    scale = float(g0.abs().max())
    fl = model._store
    lo, hi = fl.group_ranges[1][0], fl.grad_numel  # (the flow field opens the second learning-rate group)
    assert scale > 0 and float(g0[lo:hi].abs().max()) > 0
    for side in (False, True):
        for loss, g in grads[side]:
            assert abs(loss - l0) <= 1e-5 * abs(l0), (side, loss, l0)
            # float atomics order the network gradients; the tables' sums are integers (bit-equal up to the float atomics of dense levels)
            assert float((g - g0).abs().max()) <= 1e-4 * scale, (side, float((g - g0).abs().max()), scale)
            assert bool(torch.isfinite(g).all())


def test_fused_flow_loss_equals_torch_path():
    """trainer.flow_loss(fused=True) (_SceneFlowLossFn: csrc/glue.hip around the flow field's kernels, gradients straight into the
    arena) == the torch restatement of runner.py:222-253 that test_flow_loss_vs_oracle pins against the oracle: loss value and
    every flow-field parameter gradient, under an upstream gradient (loss scale) that is not 1; nothing else receives a gradient."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import flow_loss, process_pointcloud
    from oracle.detparams import fill_model
    from oracle.make_golden import SMALL_MODEL
    cfg = dict(SMALL_MODEL, num_frames=5, near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE)
    data = SyntheticKitti360(DEV, H=16, W=64, num_frames=5, num_rays=128)
    pcs, grounds = process_pointcloud(data)
    tg = torch.tensor([0.37], device=DEV)
    for frame in (2, 0, 4):  # both neighbours on both sides / only forward ones / only backward ones
        t = torch.tensor([[frame / 4]], device=DEV)
        res = {}
        for fused in (False, True):
            m = fill_model(LiDAR4D(**cfg), seed=3, flow_out_amp=0.002).to(DEV)
            m.zero_grad()
            loss = flow_loss(m, pcs, grounds, t, 5, t_ground=tg, frame_idx=frame, fused=fused)
            (loss * 512.0).backward()
            m._store.prepare_grads()
            res[fused] = (float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        (lt, gt_), (lf, gf) = res[False], res[True]
        assert abs(lt - lf) <= 2e-5 * abs(lt), (frame, lt, lf)
        for n, a in gt_.items():
            b = gf[n]
            if not n.startswith("flow_net."):
                assert float(b.abs().max()) == 0.0, n
                continue
            scale = max(float(a.abs().max()), 1e-30)
            err = float((a - b).abs().max()) / scale
            assert float(a.abs().max()) > 0 and err <= 2e-2, (frame, n, err)  # fp16 adjoints: a different power of two may be chosen
            l2 = float((a - b).norm() / a.norm())
            assert l2 <= 5e-3, (frame, n, l2)


@pytest.mark.parametrize("tag", ["default", "flow_mid", "flow_first", "flow_last", "urf", "crit_huber_bce_l1", "patch_l1", "everything"])
def test_training_losses_vs_reference_train_step_on_device(tag):
    """The reference's own train_step loss block (tests/golden/train_step_losses.npz, oracle/make_golden_train.py) against
    lidar4d_amd.trainer's terms on the device: the chamfer terms through the HIP kernel (csrc/chamfer.hip) instead of the
    oracle's brute force the fixture was generated with -- value and every gradient."""
    from tests import train_golden
    c = train_golden.load(tag)
    loss, leaves = train_golden.evaluate(c, device=DEV)
    train_golden.check(c, loss, leaves, rtol=1e-4)


def test_fused_primary_losses_vs_reference_train_step():
    """The fused primary-loss node (l4d_lidar_losses + l4d_chamfer_fwd + l4d_ray_chamfer_grad: what the benchmarked step runs)
    against the reference's train_step on the default criteria: loss value, d loss / d depth, d loss / d image."""
    from lidar4d_amd.trainer import primary_losses
    from tests import train_golden
    c = train_golden.load("default")
    o = train_golden.opt_of(c)
    depth = c["depth"].to(DEV).requires_grad_(True)
    image = c["image"].to(DEV).requires_grad_(True)
    data = {"images_lidar": c["images"].to(DEV), "rays_d_lidar": c["rays_d"].to(DEV)}
    loss = primary_losses({"depth_lidar": depth, "image_lidar": image}, data, float(o["scale"]), chamfer=True, world=1,
                          alpha_d=o["alpha_d"], alpha_r=o["alpha_r"], alpha_i=o["alpha_i"], smooth=o["smooth_factor"])
    train_golden.check(c, loss, {"g_depth": depth, "g_image": image}, rtol=1e-4)
