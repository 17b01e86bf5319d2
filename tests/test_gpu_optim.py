"""Optimiser-side semantics of the reference's AMP training loop on the HIP path (``-m gpu``, MI355X):

* torch.optim.Adam keeps its state per parameter tensor and skips tensors whose ``.grad`` is None -- the HashGridT time
  slices a step does not select (hash_field.py:79-85 under ``zero_grad(set_to_none)``): both routes (the reference-style
  loop with torch's Adam, and lidar4d_amd.trainer.FlatAdam with its device-side gates) must leave them alone;
* torch.cuda.amp.GradScaler (runner.py:102,506-508): a gradient that leaves the fp16 range must surface as inf / nan in the
  parameter gradients (the backward kernels do not saturate), the step is skipped and the scale halves; clean steps grow it;
* the reference calls render() under ``torch.autocast`` (runner.py:497);
* the parameter EMA is updated once per epoch (runner.py:534-535).
"""
import numpy as np
import pytest
import torch

from oracle.detparams import fill_model
from oracle.make_golden import SMALL_MODEL

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(num_rays=256, **kw):
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import SyntheticKitti360
    cfg = dict(SMALL_MODEL, density_scale=20.0, **kw)
    data = SyntheticKitti360(DEV, H=16, W=64, num_frames=51, num_rays=num_rays, seed=3)
    return cfg, data, (lambda seed=11: fill_model(LiDAR4D(**cfg), seed=seed).to(DEV))


def test_adam_ranges_kernel_vs_torch_adam_with_none_grads():
    """l4d_adam_step_ranges against torch.optim.Adam where a gated-off range = a tensor whose .grad is None."""
    from lidar4d_amd import ops
    torch.manual_seed(0)
    sizes, lr_mult, gate_idx = [4096, 1001, 520, 66], [1.0, 1.0, 0.1, 0.1], [-1, 0, 1, -1]
    offs = [0]
    for n in sizes[:-1]:
        offs.append(offs[-1] + (n + 7) // 8 * 8)
    total = offs[-1] + sizes[-1]
    flat = torch.randn(total, device=DEV)
    ref_params = [flat[o:o + n].clone().requires_grad_(True) for o, n in zip(offs, sizes)]
    t_opt = torch.optim.Adam([{"params": [p], "lr": 1e-2 * m} for p, m in zip(ref_params, lr_mult)], betas=(0.9, 0.99), eps=1e-15)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    f16 = torch.empty(total, dtype=torch.float16, device=DEV)
    grad = torch.zeros(total + 32, device=DEV)
    gates = grad[total:]
    steps = torch.zeros(4, dtype=torch.int32, device=DEV)
    state = torch.tensor([8.0, 0.0, 0.0, 0.125], device=DEV)  # loss scale 8: gradients arrive multiplied by 8
    ranges = ops.AdamRanges(offs, sizes, lr_mult, gate_idx)
    pattern = [(1, 0), (0, 1), (1, 1), (0, 0), (1, 0)]
    for it, (g0, g1) in enumerate(pattern):
        grad.zero_()
        gates[0], gates[1] = float(g0), float(2 * g1)  # any non-zero value opens a gate (ranks' gates are summed)
        on = [True, bool(g0), bool(g1), True]
        for p, o, n, act in zip(ref_params, offs, sizes, on):
            g = torch.randn(n, device=DEV) * (10.0 ** (it - 2))
            grad[o:o + n] = g * 8.0
            p.grad = g.clone() if act else None
        t_opt.step()
        ops.adam_step_ranges(flat, grad, m, v, f16, ranges, 1e-2, gates, state, steps, 0.9, 0.99, 1e-15)
    assert steps.tolist() == [5, 3, 2, 5]
    for p, o, n in zip(ref_params, offs, sizes):
        torch.testing.assert_close(flat[o:o + n], p.detach(), rtol=2e-6, atol=1e-7)
        assert torch.equal(f16[o:o + n], flat[o:o + n].half())
    # a raised found-inf flag skips everything, counters included
    before = flat.clone()
    state[2] = 1.0
    ops.adam_step_ranges(flat, grad, m, v, f16, ranges, 1e-2, gates, state, steps, 0.9, 0.99, 1e-15)
    assert torch.equal(flat, before) and steps.tolist() == [5, 3, 2, 5]
    ops.scaler_update(state, 2.0, 0.5, 3)
    assert state.tolist() == [4.0, 0.0, 0.0, 0.25]
    for k in range(3):
        ops.scaler_update(state, 2.0, 0.5, 3)
    assert state.tolist() == [8.0, 0.0, 0.0, 0.125]  # three clean steps -> growth
    bad = torch.zeros(1000, device=DEV)
    ops.grad_nonfinite_check(bad, state)
    assert float(state[2]) == 0.0
    bad[777] = float("nan")
    ops.grad_nonfinite_check(bad, state)
    assert float(state[2]) == 1.0


def test_untouched_time_slices_are_left_alone():
    """Frames 10, 30, 10 of 51 select the slice pairs (1, 2), (4, 5), (1, 2): slices 0, 3, 6, 7 of every HashGridT must not
    move, on the reference-style route (torch.optim.Adam: their .grad stays None -> no optimiser state) and on the FlatAdam
    route (gated ranges, per-range step counters on the device); the two routes agree on everything else."""
    from lidar4d_amd.trainer import FlatAdam, lidar_loss
    cfg, data, make = _setup()
    batches = [data.batch_for(f) for f in (10, 30, 10)]
    noises = [torch.rand(256, 64, device=DEV) for _ in batches]
    results = {}
    for route in ("torch", "flat"):
        m = make()
        init = {n: p.detach().clone() for n, p in m.named_parameters() if ".hash_t." in n}
        if route == "flat":
            m.reference_grad_none = False
            opt = FlatAdam(m, lr=1e-2, iters=100)
        else:
            opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
            sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / 100, 1))
        for b, nz in zip(batches, noises):
            opt.zero_grad()
            out = m.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, perturb=True, num_steps=64, noise=nz)
            lidar_loss(out, b["images_lidar"]).backward()
            if route == "torch":
                for n, p in m.named_parameters():
                    if ".hash_t." in n:
                        s = int(n.split(".hash_t.")[1].split(".")[0])
                        want_none = s not in ((1, 2) if b is not batches[1] else (4, 5))
                        assert (p.grad is None) == want_none, n
            opt.step()
            if route == "torch":
                sched.step()
        for n, p in m.named_parameters():
            if ".hash_t." in n:
                s = int(n.split(".hash_t.")[1].split(".")[0])
                if s in (0, 3, 6, 7):
                    assert torch.equal(p.detach(), init[n]), (route, n)
                else:
                    assert not torch.equal(p.detach(), init[n]), (route, n)
        if route == "flat":
            gated = [int(opt.steps[r]) for r, g in enumerate(opt.ranges.gate_idx) if g >= 0]
            assert gated == [0, 2, 2, 0, 1, 1, 0, 0] * 3
            assert all(int(opt.steps[r]) == 3 for r, g in enumerate(opt.ranges.gate_idx) if g < 0)
            sd = opt.state_dict()
            assert sorted({int(v["step"]) for v in sd["state"].values()}) == [1, 2, 3]
        else:
            n_params = sum(1 for g in m.get_params(1e-2) for _ in g["params"])
            assert len(opt.state_dict()["state"]) <= n_params - 4 * 3  # no state for tensors that never had a gradient
        results[route] = m._store.flat.clone()
    a, b = results["torch"], results["flat"]
    close = ((a - b).abs() <= 2e-3 * (1e-2 + a.abs())).float().mean().item()
    assert close > 0.995, close


def test_overflowing_gradient_reaches_every_parameter_group():
    """An upstream gradient far outside the fp16 range must come out as inf / nan in the gradients of every part of the
    field (sigma / attribute networks, hex-planes, static and dynamic hash tables, flow grid and flow MLP): the backward
    kernels hand non-finite values on instead of saturating them (csrc/common.h f2h_grad), which is what lets a GradScaler
    -- torch's or trainer.DynamicLossScaler -- see the overflow and skip the step."""
    from lidar4d_amd import ops
    cfg, data, make = _setup()
    m = make()
    for n_rays, T in ((32, 64), (64, 256)):  # small: atomic scatter paths; 16,384 points: sorted (binned) scatter paths
        b = data.batch_for(25)
        sel = slice(0, n_rays)
        m.zero_grad()
        out = m.render(b["rays_o_lidar"][:, sel], b["rays_d_lidar"][:, sel], b["time"], staged=False, perturb=False, num_steps=T)
        ((out["depth_lidar"].sum() + out["image_lidar"].sum()) * 1e30).backward()
        bad = {}
        for n, p in m.named_parameters():
            if p.numel() == 0 or n.startswith("unet.") or p.grad is None:
                continue
            key = n.split(".")[0] if not n.startswith("hash_encoder") else ".".join(n.split(".")[:2])
            key = "flow_net.grid" if n.startswith("flow_net.grid_enc") else ("flow_net.mlp" if n.startswith("flow_net.") else key)
            bad[key] = bad.get(key, False) or not bool(torch.isfinite(p.grad).all())
        assert set(bad) == {"planes_encoder", "hash_encoder.hash_static", "hash_encoder.hash_dynamic", "flow_net.grid",
                            "flow_net.mlp", "sigma_net", "intensity_net", "raydrop_net"}
        assert all(bad.values()), (n_rays, T, bad)
        state = torch.tensor([1.0, 0.0, 0.0, 1.0], device=DEV)
        ops.grad_nonfinite_check(m._store.flat_grad, state)
        assert float(state[2]) == 1.0
        # and an ordinary gradient stays finite
        m.zero_grad()
        out = m.render(b["rays_o_lidar"][:, sel], b["rays_d_lidar"][:, sel], b["time"], staged=False, perturb=False, num_steps=T)
        (out["depth_lidar"].sum() + out["image_lidar"].sum()).backward()
        assert bool(torch.isfinite(m._store.flat_grad).all())


def test_dynamic_loss_scaler_skips_backs_off_and_grows():
    """trainer.DynamicLossScaler inside Trainer.train_step: from an absurd initial scale every step overflows, is skipped
    (parameters, moments and step counters untouched) and halves the scale, until the gradients fit; from then on the
    steps are applied and, with a short growth interval, the scale doubles again."""
    from lidar4d_amd.trainer import Trainer
    cfg, data, make = _setup(num_rays=128)
    m = make()
    tr = Trainer(m, data, num_steps=64, iters=100, chamfer=False, flow=False, init_scale=2.0 ** 60)
    tr.scaler.growth_interval = 4
    init = m._store.flat.clone()
    b = data.batch_for(20)
    scales, moved_at = [], None
    for it in range(64):
        loss = tr.train_step(b)
        assert np.isfinite(float(loss))
        scales.append(tr.scaler.get_scale())
        moved = not torch.equal(m._store.flat, init)
        if moved and moved_at is None:
            moved_at = it
            assert int(tr.opt.steps.max()) == 1
        if moved_at is None:
            assert int(tr.opt.steps.max()) == 0 and float(tr.opt.exp_avg.abs().max()) == 0.0
            assert scales[-1] == 2.0 ** (59 - it)
        if moved_at is not None and it >= moved_at + 12:
            break
    assert moved_at is not None and moved_at >= 10, (moved_at, scales[:8])
    assert bool(torch.isfinite(m._store.flat).all())
    assert max(scales[moved_at:]) > scales[moved_at], "four clean steps in a row double the scale"
    assert tr.opt.step_count == len(scales)  # the lr schedule advances on skipped steps too (runner.py:510-511)


def test_render_under_autocast_with_torch_gradscaler():
    """runner.py:497-508 as written: autocast(fp16) around render + loss (+ the U-Net), scaler.scale(loss).backward(),
    scaler.step(optimizer), scaler.update() -- with torch's own Adam and GradScaler on the HIP model.  The fused node opts
    out of autocast (its precisions are fixed), so the result equals the un-autocast call; an absurd scale is backed off."""
    from lidar4d_amd.trainer import lidar_loss
    cfg, data, make = _setup()
    b = data.batch_for(25)
    noise = torch.rand(256, 64, device=DEV)
    m = make()
    m.unet.eval()
    with torch.no_grad():
        plain = m.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, perturb=True, num_steps=64, noise=noise)
    m.loss_scale = 1.0  # the outer GradScaler provides the fp16 loss scale
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40, growth_interval=10 ** 9)
    before = m._store.flat.clone()
    applied = 0
    for it in range(40):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            out = m.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, perturb=True, num_steps=64, noise=noise)
            assert out["depth_lidar"].dtype == torch.float32 and out["image_lidar"].dtype == torch.float32
            loss = lidar_loss(out, b["images_lidar"])
            img = out["image_lidar"].reshape(1, 16, 16, 2)
            stacked = torch.cat([img[..., 0], img[..., 1], out["depth_lidar"].reshape(1, 16, 16)], 0).unsqueeze(0)
            with torch.no_grad():  # the refinement network is evaluated under autocast too (runner.py:413-416: fp16 convs)
                assert bool(torch.isfinite(m.unet(stacked.detach())).all())
        if it == 0:
            assert torch.equal(out["depth_lidar"], plain["depth_lidar"]) and torch.equal(out["image_lidar"], plain["image_lidar"])
        scaler.scale(loss).backward()
        scale_before = scaler.get_scale()
        scaler.step(opt)
        scaler.update()
        if scaler.get_scale() == scale_before:
            applied += 1
            if applied == 2:
                break
        else:
            assert torch.equal(m._store.flat, before), "an overflowing step must not touch the parameters"
    assert applied == 2 and scaler.get_scale() < 2.0 ** 40
    assert bool(torch.isfinite(m._store.flat).all()) and not torch.equal(m._store.flat, before)


def test_ema_updates_once_per_epoch():
    from lidar4d_amd.trainer import Trainer
    cfg, data, make = _setup(num_rays=64)
    m = make()
    tr = Trainer(m, data, num_steps=32, iters=100, chamfer=False, flow=False, ema_decay=0.95, epoch_steps=3, init_scale=1.0)
    b = data.batch_for(5)
    for _ in range(7):
        tr.train_step(b)
    assert tr.ema.num_updates == 2
    assert Trainer(make(), data, num_steps=32, chamfer=False, flow=False, ema_decay=0.95).epoch_steps == data.num_frames


def test_flow_loss_overflow_is_skipped_not_clipped():
    """with the scene-flow loss on, an overflowing loss scale must reach the flow field's gradients as inf / nan
    through ``model.flow()`` (flow_field.py backward no longer clamps to +-65504) so that the step is skipped and the scale
    backs off -- not clipped silently while the scale keeps growing."""
    from lidar4d_amd.trainer import Trainer
    cfg, data, make = _setup(num_rays=128)
    m = make()
    tr = Trainer(m, data, num_steps=64, iters=100, chamfer=False, flow=True, init_scale=2.0 ** 40)
    init = m._store.flat.clone()
    scales = []
    for it in range(6):
        loss = tr.train_step(data.batch_for(20))
        assert np.isfinite(float(loss))
        scales.append(tr.scaler.get_scale())
    assert scales == [2.0 ** (39 - i) for i in range(6)], scales   # every step overflowed and halved the scale
    assert torch.equal(m._store.flat, init) and int(tr.opt.steps.max()) == 0
    # the flow field's own adjoint (model.flow() outside the fused node) normalises its fp16 range on the device from the
    # upstream gradient it receives: whatever power of two the caller scaled the loss by comes back out exactly ...
    from lidar4d_amd.trainer import flow_loss
    tg = torch.tensor([0.37], device=DEV)
    grads = []
    for s in (1.0, 2.0 ** 40):
        tr.opt.zero_grad()
        fl = flow_loss(m, tr.pc_list, tr.pc_ground_list, data.batch_for(20)["time"], data.num_frames, t_ground=tg, frame_idx=20)
        (fl * s).backward()
        grads.append([p.grad.detach().clone() for p in m.flow_net.parameters()])
    for g1, g40 in zip(*grads):
        assert bool(torch.isfinite(g40).all()) and float(g1.abs().max()) > 0
        # (bit-equal up to the order of the dW atomics: compare at 1e-4 of the tensor's largest gradient)
        err = float((g1.double() * 2.0 ** 40 - g40.double()).abs().max()) / (float(g1.abs().max()) * 2.0 ** 40)
        assert err < 1e-4, "the flow adjoint must be invariant under power-of-two loss scales (%.2e)" % err
    # ... and an upstream gradient that is itself non-finite surfaces as inf / nan (never as a clipped number)
    tr.opt.zero_grad()
    fl = flow_loss(m, tr.pc_list, tr.pc_ground_list, data.batch_for(20)["time"], data.num_frames, t_ground=tg, frame_idx=20)
    (fl * 3.0e38).backward()
    g = m.flow_net.grid_enc.params.grad
    assert not bool(torch.isfinite(g).all()), "an overflowing flow-loss gradient must surface as inf / nan, not as a clipped number"


def test_graphed_step_replays_the_training_step():
    """Trainer.train_step_graphed: one hipGraph per frame index holding the whole step (batch draw, forward, losses, backward,
    scaler, Adam with the learning-rate schedule on the device), on the default model at the reference's own batch of 1,024
    rays.  Replays must keep training (parameters move, gradients finite, step counters advance), draw NEW rays every replay
    (the dataset's device generator is registered with the graph), and advance the device-side schedule like the host's."""
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import Trainer
    torch.manual_seed(0)
    m = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(DEV)
    data = SyntheticKitti360(DEV, W=1024, num_rays=1024, seed=5, frame_seed=5)
    tr = Trainer(m, data, iters=200, chamfer=True, flow=True, ema_decay=None)
    assert tr.graphs_supported()
    for _ in range(12):  # let the loss scale settle (it backs off from 65536 while the gradients overflow)
        tr.train_step(data.batch_for(20))
    n0 = int(tr.opt.steps[0])
    assert n0 >= 1
    losses = []
    for it in range(8):
        frame = (20, 21)[it % 2]
        before = m._store.flat.clone()
        loss = tr.train_step_graphed(frame)
        losses.append(float(loss))
        assert np.isfinite(losses[-1])
        assert bool(torch.isfinite(m._store.flat_grad).all()), f"step {it}: non-finite gradients"
        assert not torch.equal(m._store.flat, before), f"step {it} ({'capture call' if it < 2 else 'replay'}) did not move the parameters"
    assert len(tr._step_graphs["graphs"]) == 2
    assert int(tr.opt.steps[0]) == n0 + 8
    # replays of the same frame's graph see different batches: the loss values differ from replay to replay
    assert len({round(v, 6) for v in losses[2::2]}) > 1
    assert tr.opt.step_count == 20
    sched = tr.opt.sched.tolist()
    assert sched[0] == 20.0 and abs(sched[1] - 0.1 ** (19 / 200)) < 1e-6  # factor of the LAST step = 0.1 ** min(19 / iters, 1)
    before = m._store.flat.clone()  # an eager step still works afterwards (shared device state, host-side caches consistent)
    assert np.isfinite(float(tr.train_step(data.batch_for(22))))
    assert not torch.equal(m._store.flat, before) and tr.opt.sched.tolist()[0] == 21.0


@pytest.mark.parametrize("streams", [0, 2])
def test_graph_replay_equals_eager_step(streams, monkeypatch):
    """A REPLAY of the captured step must produce the gradients and the parameter update the eager step produces from the same
    state and the same batch -- also from the second replay on, which is where memset nodes in the graph (a hipMemsetAsync fill,
    torch's multi-block reduce zeroing its semaphores) made single-chain graphs go wrong on ROCm 7.2 (DESIGN.md section 5,
    tools/graph_diff.py): the first replay was exact, every later one differed in the time planes' gradients.  Random draws are
    pinned (static batch, no sample jitter, fixed ground-flow time); tolerance = the order of the dW atomics."""
    from lidar4d_amd import LiDAR4D, _lib, ops, trainer as trainer_mod
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.params import bump_epoch
    from lidar4d_amd.trainer import Trainer
    mask_was = ops.streams_mask()
    _lib.lib().l4d_streams_config(streams)
    try:
        torch.manual_seed(0)
        m = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(DEV)
        data = SyntheticKitti360(DEV, W=1024, num_rays=1024, seed=7, frame_seed=7)
        tr = Trainer(m, data, iters=200, chamfer=True, flow=True, ema_decay=None, init_scale=1024.0, graph_batch_inside=False)
        st, opt = m._store, tr.opt
        batch = {k: (v.contiguous().clone() if torch.is_tensor(v) else v) for k, v in data.batch_for(20).items()}
        monkeypatch.setattr(data, "batch_for", lambda frame: batch)
        render = m.render
        monkeypatch.setattr(m, "render", lambda *a, **kw: render(*a, **{**kw, "perturb": False}))
        tg, fl = torch.tensor([0.37], device=DEV), trainer_mod.flow_loss
        monkeypatch.setattr(trainer_mod, "flow_loss", lambda *a, **kw: fl(*a, **{**kw, "t_ground": tg}))
        for _ in range(4):
            tr.train_step(batch)
        opt.device_schedule()
        snap = {"flat": st.flat.detach().clone(), "m": opt.exp_avg.clone(), "v": opt.exp_avg_sq.clone(), "steps": opt.steps.clone(),
                "scaler": tr.scaler.state.clone(), "sched": opt.sched.clone(), "count": opt.step_count}

        def restore():
            with torch.no_grad():
                st.flat.copy_(snap["flat"]), opt.exp_avg.copy_(snap["m"]), opt.exp_avg_sq.copy_(snap["v"]), opt.steps.copy_(snap["steps"])
                tr.scaler.state.copy_(snap["scaler"]), opt.sched.copy_(snap["sched"])
            opt.step_count = snap["count"]
            bump_epoch()
            st.refresh16()

        restore()
        tr.train_step(batch)
        g_e, p_e = st.flat_grad.detach().clone(), st.flat.detach().clone()
        assert bool(torch.isfinite(g_e).all()) and not torch.equal(p_e, snap["flat"])
        restore()
        tr.train_step_graphed(20)  # eager warm-up + capture
        for k in range(4):
            restore()
            tr.train_step_graphed(20)
            for name, p, off, n, gi in st.entries:
                if not n:
                    continue
                a, b = st.flat_grad[off:off + n], g_e[off:off + n]
                assert bool(torch.isfinite(a).all()), f"replay {k}: non-finite gradient in {name}"
                d = float((a.double() - b.double()).abs().max()) / max(float(b.abs().max()), 1e-30)
                assert d < 1e-3, f"replay {k}: gradient of {name} differs from the eager step's by {d:.2e} of its largest value"
            # (Adam with eps = 1e-15 turns the sign of a rounding-noise gradient into a +-lr step: a handful of parameters may differ)
            off_frac = float(((st.flat - p_e).abs() > 1e-4).float().mean())
            assert off_frac < 1e-4, f"replay {k}: {off_frac:.2e} of the parameters differ from the eager step's"
    finally:
        _lib.lib().l4d_streams_config(mask_was)


def test_device_schedule_matches_host_schedule():
    """l4d_adam_step_ranges with the learning-rate schedule on the device (``sched``) against the same launch with the
    host-computed rate: the factor 0.1 ** min(it / iters, 1) of every iteration, and the parameters after each step (the rate
    is rounded to fp32 once in either route: 1e-6 relative).  (Whole training runs are no basis for this comparison: with
    eps = 1e-15 Adam turns the sign of a rounding-noise gradient into a full +-lr step.)"""
    from lidar4d_amd import ops
    torch.manual_seed(3)
    n, iters, lr0 = 4096, 5, 1e-2
    ranges = ops.AdamRanges([0, 2048], [2048, 2048], [1.0, 0.1], [-1, -1])
    mk = lambda: (torch.randn(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV),
                  torch.empty(n, dtype=torch.float16, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))
    pa, ma, va, ha, sa = mk()
    pb, mb, vb, hb, sb = pa.clone(), ma.clone(), va.clone(), ha.clone(), sa.clone()
    gates = torch.zeros(4, device=DEV)
    sched = torch.tensor([0.0, 1.0], device=DEV)
    for it in range(8):  # runs past iters: the factor saturates at 0.1
        g = torch.randn(n, device=DEV)
        ops.adam_step_ranges(pa, g, ma, va, ha, ranges, lr0 * 0.1 ** min(it / iters, 1.0), gates, None, sa, 0.9, 0.99, 1e-15)
        ops.adam_step_ranges(pb, g, mb, vb, hb, ranges, lr0, gates, None, sb, 0.9, 0.99, 1e-15, sched=sched, sched_iters=float(iters))
        it_dev, factor = sched.tolist()
        assert it_dev == it + 1 and abs(factor - 0.1 ** min(it / iters, 1.0)) < 1e-7
        assert float((pa - pb).abs().max()) <= 2e-6 * float(pa.abs().max()), it
    assert torch.equal(sa, sb) and float((ma - mb).abs().max()) == 0.0
