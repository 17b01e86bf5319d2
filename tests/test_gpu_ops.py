"""GPU parity tests, operator level: every HIP entry point (called through the C ABI via lidar4d_amd.ops)
against the oracle with tiny-cuda-nn's rounding points, on seeded inputs.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch

from oracle import fields_ref, tcnn_ref
from oracle.detparams import det_uniform

pytestmark = pytest.mark.gpu

DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_close(got, ref, rtol, atol, what="", frac_ok=0.0):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    bad = (got - ref).abs() > atol + rtol * ref.abs()
    nbad = int(bad.sum())
    assert nbad <= frac_ok * bad.numel(), (
        f"{what}: {nbad}/{bad.numel()} out of tolerance; max abs err {(got - ref).abs().max():.3e}, ref max {ref.abs().max():.3e}")


HALF_ULP = 2.0 ** -10  # one fp16 ulp, relative


@pytest.fixture(autouse=True)
def _tcnn_mode():
    prev = tcnn_ref.get_precision()
    tcnn_ref.set_precision("tcnn")
    yield
    tcnn_ref.set_precision(prev)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,F,L,log2T,base,maxres", [(3, 4, 8, 19, 512, 32768), (2, 4, 8, 13, 512, 32768),
                                                      (3, 8, 8, 18, 32, 8192), (3, 2, 4, 10, 4, 32), (2, 8, 3, 8, 8, 64),
                                                      (3, 4, 16, 15, 512, 32768)])
def test_hashgrid_fwd_bwd(D, F, L, log2T, base, maxres):
    from lidar4d_amd import tcnn
    cfg = {"otype": "HashGrid", "n_levels": L, "n_features_per_level": F, "log2_hashmap_size": log2T,
           "base_resolution": base, "per_level_scale": np.exp2(np.log2(maxres / base) / (L - 1))}
    ref = tcnn_ref.Encoding(D, cfg)
    mod = tcnn.Encoding(D, cfg)
    assert mod.params.numel() == ref.params.numel()
    assert mod.meta.res == ref.meta["res"] and mod.meta.size == ref.meta["size"] and mod.meta.hashed == ref.meta["hashed"]
    with torch.no_grad():
        ref.params.copy_(det_uniform((ref.params.numel(),), f"hg{D}{F}{L}", -0.5, 0.5))
        mod.params.copy_(ref.params)
    mod = mod.to(DEV)
    P = 20000
    x = det_uniform((P, D), f"x{D}{F}", -0.02, 1.02)  # a few points outside [0,1]: indices wrap like tiny-cuda-nn
    x[:50] = x[:50].clamp(0, 1).round()
    out_ref = ref(x)
    out = mod(x.to(DEV))
    assert out.dtype == torch.float16 and out.shape == (P, L * F)
    # identical rounding points; fp32 accumulation order may differ -> at most one fp16 ulp, rarely
    rel_close(out.float(), out_ref.float(), rtol=2 * HALF_ULP, atol=1e-6, what="hashgrid fwd")
    exact = (out.cpu() == out_ref).float().mean().item()
    assert exact > 0.98, f"only {exact:.4f} of outputs bit-identical"
    g = det_uniform((P, L * F), "g", -1, 1)
    out_ref.float().backward(g)
    out.backward(g.to(DEV).half())
    rel_close(mod.params.grad, ref.params.grad, rtol=2e-3, atol=2e-3 * ref.params.grad.abs().max().item(), what="hashgrid bwd")


@pytest.mark.parametrize("t", [0.0, 0.3, 1.0, 26 / 51, 0.62])
def test_hashgrid_t(t):
    from lidar4d_amd.hash_field import HashGridT
    ref = fields_ref.HashGridT(time_resolution=8, base_resolution=512, max_resolution=32768, n_levels=8,
                               n_features_per_level=4, log2_hashmap_size=13)
    mod = HashGridT(time_resolution=8, base_resolution=512, max_resolution=32768, n_levels=8, n_features_per_level=4,
                    log2_hashmap_size=13)
    with torch.no_grad():
        for (n, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
            p.copy_(det_uniform(tuple(p.shape), "ht" + n, -0.5, 0.5))
            q.copy_(p)
    mod = mod.to(DEV)
    x = det_uniform((8192, 2), "htx", 0, 1)
    tt = torch.tensor(t, dtype=torch.float32)
    out_ref = ref(x, tt)
    out = mod(x.to(DEV), tt)
    assert out.dtype == torch.float32 and out.shape == (8192, 8)
    rel_close(out, out_ref, rtol=1e-4, atol=3e-4, what="hashgrid_t fwd")  # fp16-ulp flips of slice features, scaled by basis
    g = det_uniform((8192, 8), "htg", -1, 1)
    out_ref.backward(g)
    out.backward(g.to(DEV))
    for (n, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().sum()) == 0.0, n
        else:
            # fp32 atomics accumulate thousands of terms per entry in arbitrary order
            rel_close(q.grad, p.grad, rtol=1e-3, atol=5e-4 * max(p.grad.abs().max().item(), 1e-9), what="hashgrid_t bwd " + n)


@pytest.mark.parametrize("F,num_basis,t", [(8, 4, 0.3), (4, 2, 0.62), (8, 2, 1.0), (2, 2, 0.3)])
def test_hashgrid_t_other_widths(F, num_basis, t):
    """HashGridT with feature widths / basis counts off the reference default (hash_field.py:39,65-74): slice by slice through the
    generic hash-grid kernels, blend and interpT as torch arithmetic -- against the oracle, forward and table gradients."""
    from lidar4d_amd.hash_field import HashGridT
    kw = dict(time_resolution=8, base_resolution=16, max_resolution=512, n_levels=8, n_features_per_level=F, log2_hashmap_size=11,
              num_basis=num_basis)
    ref, mod = fields_ref.HashGridT(**kw), HashGridT(**kw)
    assert not mod.fused and mod.n_output_dims == ref.n_output_dims == 8 * F // num_basis
    with torch.no_grad():
        for (n, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
            p.copy_(det_uniform(tuple(p.shape), "hw" + n, -0.5, 0.5))
            q.copy_(p)
    mod = mod.to(DEV)
    x, tt = det_uniform((4096, 2), "hwx", 0, 1), torch.tensor(t, dtype=torch.float32)
    out_ref, out = ref(x, tt), mod(x.to(DEV), tt)
    assert out.shape == (4096, ref.n_output_dims)
    rel_close(out.float(), out_ref.float(), rtol=1e-3, atol=1e-3, what="hashgrid_t (composed) fwd")
    g = det_uniform(tuple(out_ref.shape), "hwg", -1, 1)
    out_ref.backward(g)
    out.backward(g.to(DEV).to(out.dtype))
    for (n, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().sum()) == 0.0, n
        else:
            rel_close(q.grad, p.grad, rtol=3e-3, atol=3e-3 * max(p.grad.abs().max().item(), 1e-9), what="hashgrid_t (composed) bwd " + n)
    with pytest.raises(ValueError):
        HashGridT(n_features_per_level=4, num_basis=3)


@pytest.mark.parametrize("reduction,decompose", [("sum", True), ("prod", False), ("mean", True), ("concat", False)])
def test_hashgrid4d_reduction_and_decompose(reduction, decompose):
    """HashGrid4D's non-default options (hash_field.py:16-27,101-102,134-138,155-170): the three 2-D x time stacks combined by
    product / sum / mean instead of concatenation, and one concatenated tensor instead of the [static, dynamic] pair --
    operator-level path, against the oracle's restatement, forward and parameter gradients."""
    from lidar4d_amd.hash_field import HashGrid4D
    kw = dict(base_resolution=16, max_resolution=256, time_resolution=8, n_levels=4, n_features_per_level=4, log2_hashmap_size=12,
              hash_size_dynamic=(10, 9, 9), decompose=decompose, reduction=reduction)
    ref, mod = fields_ref.HashGrid4D(**kw), HashGrid4D(**kw)
    assert mod.n_output_dims == ref.n_output_dims == 16 + (12 if reduction == "concat" else 4)
    with torch.no_grad():
        for (n, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
            p.copy_(det_uniform(tuple(p.shape), "h4" + n, -0.5, 0.5))
            q.copy_(p)
    mod = mod.to(DEV)
    x, t = det_uniform((4096, 3), "h4x", 0, 1), torch.tensor(0.3, dtype=torch.float32)
    o_ref, o = ref(x, t), mod(x.to(DEV), t)
    if decompose:
        assert isinstance(o, list) and len(o) == 2
        o_ref, o = torch.cat([v.float() for v in o_ref], -1), torch.cat([v.float() for v in o], -1)
    assert o.shape == (4096, ref.n_output_dims)
    rel_close(o.float(), o_ref.float(), rtol=2e-3, atol=1e-3, what=f"HashGrid4D {reduction} fwd")
    g = det_uniform(tuple(o_ref.shape), "h4g", -1, 1)
    o_ref.float().backward(g)
    o.float().backward(g.to(DEV))
    for (n, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().sum()) == 0.0, n
        else:
            rel_close(q.grad, p.grad, rtol=3e-3, atol=3e-3 * max(p.grad.abs().max().item(), 1e-9), what=f"HashGrid4D {reduction} bwd " + n)
    with pytest.raises(ValueError):
        HashGrid4D(reduction="max")


def test_planes_vs_reference_golden(golden):
    """Planes4D against fixtures produced by the reference's own F.grid_sample code (make_golden.py (3))."""
    from lidar4d_amd.planes_field import Planes4D
    g = golden("planes4d")
    mod = Planes4D(output_dim=8, resolution=(8, 8, 8, 8), multiscale_res=(1, 2, 4))
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_(T(g["param." + n]))
    mod = mod.to(DEV)
    xt = T(g["xt"]).to(DEV).requires_grad_(True)
    fs, fd = mod(xt)
    rel_close(fs, T(g["feat_static"]), 2e-5, 1e-6, "planes static")
    rel_close(fd, T(g["feat_dynamic"]), 2e-5, 1e-6, "planes dynamic")
    rel_close(mod.forward_static(xt), T(g["feat_static_only"]), 2e-5, 1e-6, "planes static-only")
    rel_close(mod.forward_dynamic(xt), T(g["feat_dynamic_only"]), 2e-5, 1e-6, "planes dynamic-only")
    ((fs * T(g["gs"]).to(DEV)).sum() + (fd * T(g["gd"]).to(DEV)).sum()).backward()
    rel_close(xt.grad, T(g["grad_xt"]), 1e-4, 1e-4, "planes d/dxt")
    for n, p in mod.named_parameters():
        rel_close(p.grad, T(g["grad." + n]), 1e-4, 1e-4, "planes grad " + n)


def test_frequency():
    from lidar4d_amd import tcnn
    ref = tcnn_ref.Encoding(3, {"otype": "Frequency", "degree": 12})
    mod = tcnn.Encoding(3, {"otype": "Frequency", "degree": 12})
    x = det_uniform((4096, 3), "fx", 0, 1)
    x[:8] = torch.tensor([[0.0, 0.5, 1.0]] * 8)
    out = mod(x.to(DEV))
    assert out.shape == (4096, 72) and out.dtype == torch.float16
    rel_close(out.float(), ref(x).float(), rtol=2 * HALF_ULP, atol=2e-6, what="frequency")


@pytest.mark.parametrize("n_in,n_out,n_hidden", [(120, 16, 1), (87, 1, 2), (120, 16, 2), (16, 6, 2), (60, 3, 3), (176, 16, 2), (150, 16, 1)])
def test_mlp_fwd_bwd(n_in, n_out, n_hidden):
    from lidar4d_amd import tcnn
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
           "n_hidden_layers": n_hidden}
    ref = tcnn_ref.Network(n_in, n_out, cfg)
    mod = tcnn.Network(n_in, n_out, cfg)
    with torch.no_grad():
        ref.params.copy_(det_uniform((ref.params.numel(),), f"mlp{n_in}", -0.3, 0.3))
        mod.params.copy_(ref.params)
    mod = mod.to(DEV)
    P = 5000  # not a multiple of 16/32: exercises the ragged tail
    x = det_uniform((P, n_in), "mx", -1, 1).requires_grad_(True)
    xg = x.detach().to(DEV).requires_grad_(True)
    y_ref = ref(x)
    y = mod(xg)
    assert y.dtype == torch.float16 and y.shape == (P, n_out)
    rel_close(y.float(), y_ref.float(), rtol=4 * HALF_ULP, atol=2e-3, what="mlp fwd", frac_ok=1e-3)
    g = det_uniform((P, n_out), "mg", -1, 1)
    y_ref.float().backward(g)
    y.backward(g.to(DEV).half())
    gmax = ref.params.grad.abs().max().item()
    rel_close(mod.params.grad, ref.params.grad, rtol=2e-2, atol=2e-3 * gmax, what="mlp dW")
    rel_close(xg.grad, x.grad, rtol=2e-2, atol=2e-3 * x.grad.abs().max().item(), what="mlp dX")


@pytest.mark.parametrize("in_pad,n_hidden", [(16, 2), (16, 3), (32, 1)])
def test_mlp_bwd_recomputed_activations_equal_saved(in_pad, n_hidden):
    """l4d_mlp_bwd with act = null recomputes the hidden activations from x with the forward chain: the same MFMAs in the
    same order as the forward, so dx is bit-identical to the saved-activation path and dW differs by atomics order only."""
    from lidar4d_amd import ops
    assert ops.mlp_recompute_supported(in_pad, n_hidden) and not ops.mlp_recompute_supported(96, 2)
    P = 70001  # ragged tail
    w = (det_uniform((64 * in_pad + (n_hidden - 1) * 64 * 64 + 16 * 64,), f"rw{in_pad}", -0.3, 0.3)).half().to(DEV)
    x = det_uniform((P, in_pad), "rx", -1, 1).half().to(DEV)
    dy = det_uniform((P, 16), "rdy", -1, 1).half().to(DEV)
    n = torch.tensor([P - 37], dtype=torch.int32, device=DEV)
    y, act = ops.mlp_fwd(x, w, n_hidden, save_act=True, n_rows=n)
    y2, none = ops.mlp_fwd(x, w, n_hidden, save_act=False, n_rows=n)
    assert none is None and torch.equal(y[:P - 37], y2[:P - 37])
    g1, g2 = torch.zeros(w.numel(), device=DEV), torch.zeros(w.numel(), device=DEV)
    dx1 = ops.mlp_bwd(x, act, dy, w, n_hidden, g1, 1.0 / 128, n_rows=n)
    dx2 = ops.mlp_bwd(x, None, dy, w, n_hidden, g2, 1.0 / 128, n_rows=n)
    assert torch.equal(dx1[:P - 37], dx2[:P - 37])
    assert float((g1 - g2).abs().max()) <= 1e-5 * float(g1.abs().max()) and float(g1.abs().max()) > 0
    with pytest.raises(Exception):
        ops.mlp_bwd(x[:, :16].contiguous().repeat(1, 6), None, dy, torch.zeros(64 * 96 + 64 * 64 + 16 * 64, device=DEV).half(), 2,
                    torch.zeros(64 * 96 + 64 * 64 + 16 * 64, device=DEV), 1.0)


def test_mlp_empty_and_row_count():
    from lidar4d_amd import ops
    w = torch.randn(64 * 96 + 64 * 64 + 16 * 64, device=DEV).half() * 0.1
    x = torch.randn(1000, 96, device=DEV).half()
    y_all, _ = ops.mlp_fwd(x, w, 2, save_act=False)
    n = torch.tensor([333], dtype=torch.int32, device=DEV)
    y = torch.full((1000, 16), 7.0, dtype=torch.float16, device=DEV)
    ops.mlp_fwd(x, w, 2, save_act=False, n_rows=n, y=y)
    assert torch.equal(y[:333], y_all[:333]) and bool((y[333:] == 7.0).all())
    y0, _ = ops.mlp_fwd(x[:0], w, 2, save_act=False)
    assert y0.shape == (0, 16)


@pytest.mark.parametrize("tag", ["plain", "active_perturb", "tightbound"])
def test_sample_and_composite_vs_reference_golden(golden, tag):
    """l4d_sample_rays + l4d_composite_fwd against LiDAR_Renderer.run of the reference (make_golden.py (2))."""
    from lidar4d_amd import ops
    g = golden("run_analytic_" + tag)
    ro, rd = T(g["rays_o"]).view(-1, 3).to(DEV), T(g["rays_d"]).view(-1, 3).to(DEV)
    lin = torch.linspace(0.0, 1.0, 768).to(DEV)  # the reference's torch.linspace bits
    noise = T(g["noise"]).to(DEV) if bool(g["perturb"]) else None
    z, xyz = ops.sample_rays(ro, rd, lin, noise, float(g["near"]), float(g["far"]), float(g["bound"]))
    assert torch.equal(z.cpu(), T(g["z_vals"])), "z_vals must be bit-exact"
    r2 = ((xyz.cpu() - torch.tensor([0.2, 0.1, -0.05])) ** 2).sum(-1)
    sigma = (400.0 * torch.exp(-r2 / 0.02) + 0.3).view(-1, 768).to(DEV)  # the fixture's analytic density, CPU torch
    sd = float(np.float32(np.float32(g["far"]) - np.float32(g["near"])) / np.float32(768))
    w, wsum, depth, mask, idx, cnt = ops.composite_fwd(sigma, z, sd, float(g["density_scale"]), bool(g["active"]))
    rel_close(w, T(g["weights"]), 2e-5, 3e-7, "weights")  # alpha = 1 - exp(.): one ulp(1) absolute
    rel_close(depth, T(g["depth"]).view(-1), 1e-4, 1e-8, "depth")
    rel_close(wsum, T(g["weights_sum"]), 1e-4, 1e-8, "weights_sum")
    want = set(g["mask_idx"].tolist())
    got_mask = set(torch.nonzero(mask.view(-1)).view(-1).tolist())
    got_idx = set(idx[: int(cnt)].tolist())
    assert got_mask == got_idx, "compacted index list != dense mask"
    wref = T(g["weights"]).reshape(-1)
    for i in got_mask ^ want:
        assert abs(float(wref[i]) - 1e-4) < 2e-9, (i, float(wref[i]))
    xn = xyz.cpu()
    geo = torch.stack([xn[:, 0], xn[:, 1] * 2], -1)
    dirs = rd.cpu().view(-1, 1, 3).expand(-1, 768, 3).reshape(-1, 3)
    a = torch.stack([torch.sigmoid(geo[:, 0] * 3 + dirs[:, 0]), torch.sigmoid(geo[:, 1] - dirs[:, 2])], -1)
    attr = torch.zeros_like(a)
    m = mask.view(-1).bool().cpu()
    attr[m] = a[m]
    image = ops.composite_image(w, attr.to(DEV).contiguous(), 2)
    rel_close(image, T(g["image"]).view(-1, 2), 1e-4, 1e-8, "image")


@pytest.mark.parametrize("active,T_steps", [(False, 768), (True, 100), (False, 1500)])
def test_composite_bwd(active, T_steps):
    from lidar4d_amd import ops
    N = 37
    sigma = (det_uniform((N, T_steps), "cs", 0, 1) ** 4 * 300).requires_grad_(True)
    z, sd = fields_ref.sample_z(N, 0.0105, 0.851, T_steps, det_uniform((N, T_steps), "cn", 0, 1))
    attr = det_uniform((N * T_steps, 2), "ca", 0, 1).requires_grad_(True)
    w = fields_ref.composite(sigma, z, sd, 1.3, active)
    depth, wsum, image = (w * z).sum(-1), w.sum(-1), (w.unsqueeze(-1) * attr.view(N, T_steps, 2)).sum(-2)
    gd, gs, gi, gw = (det_uniform((N,), "gd", -1, 1), det_uniform((N,), "gs", -1, 1), det_uniform((N, 2), "gi", -1, 1),
                      det_uniform((N, T_steps), "gw", -1, 1))
    ((depth * gd).sum() + (wsum * gs).sum() + (image * gi).sum() + (w * gw).sum()).backward()
    sdv = float(sd.reshape(-1)[0])
    wg, _, _, _, _, _ = ops.composite_fwd(sigma.detach().to(DEV), z.to(DEV).contiguous(), sdv, 1.3, active)
    rel_close(wg, w, 3e-5, 3e-7, "weights")  # alpha = 1 - exp(.) carries one ulp(1) = 6e-8 of absolute error
    d_sigma, d_attr = ops.composite_bwd(sigma.detach().to(DEV), z.to(DEV).contiguous(), wg, attr.detach().to(DEV), 2, sdv, 1.3,
                                        active, gd.to(DEV), gs.to(DEV), gi.to(DEV), gw.to(DEV))
    rel_close(d_sigma, sigma.grad, 2e-3, 1e-6 * sigma.grad.abs().max().item(), "d_sigma")
    rel_close(d_attr, attr.grad, 1e-4, 3e-7, "d_attr")


def test_adam_matches_torch():
    from lidar4d_amd import ops
    n = 100003
    p = torch.randn(n, device=DEV)
    p_ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p16 = torch.empty(n, dtype=torch.float16, device=DEV)
    for step in range(1, 4):
        g = torch.randn(n, device=DEV)
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g, m, v, p16, 1e-2, 0.9, 0.99, 1e-15, step)
        rel_close(p, p_ref, 1e-5, 1e-6, f"adam step {step}")
    assert torch.equal(p16, p.half())


@pytest.mark.parametrize("n", [0, 1, 7, 4096, 1000003])
def test_absmax(n):
    """l4d_absmax_f32: exactly torch's x.abs().max() (a maximum has no rounding); +inf as soon as one value is inf / nan."""
    from lidar4d_amd import ops
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, generator=g) * 37.0).to(DEV)
    got = ops.absmax(x)
    assert got.shape == (1,) and got.dtype == torch.float32
    assert float(got) == (float(x.abs().max()) if n else 0.0)
    if n:
        for bad in (float("inf"), float("-inf"), float("nan")):
            y = x.clone()
            y[n // 2] = bad
            assert float(ops.absmax(y)) == float("inf")


@pytest.mark.parametrize("b,n,m", [(1, 1000, 1700), (2, 513, 257), (1, 16384, 16384)])
def test_chamfer_fwd_bwd(b, n, m):
    """l4d_chamfer_fwd/bwd (replacement of utils/chamfer3D/chamfer3D.cu) against the brute-force oracle."""
    from lidar4d_amd.chamfer import chamfer_3DDist
    from oracle.chamfer_ref import chamfer as chamfer_ref
    g = torch.Generator().manual_seed(b * 1000 + n)
    a = (torch.rand(b, n, 3, generator=g) * 2 - 1).requires_grad_(True)
    c = (torch.rand(b, m, 3, generator=g) * 2 - 1).requires_grad_(True)
    ag, cg = a.detach().to(DEV).requires_grad_(True), c.detach().to(DEV).requires_grad_(True)
    d1, d2, i1, i2 = chamfer_3DDist()(ag, cg)
    assert d1.shape == (b, n) and d2.shape == (b, m) and i1.dtype == torch.int32
    if n * m <= 4_000_000:
        r1, r2, j1, j2 = chamfer_ref(a, c)
        assert torch.equal(d1.cpu(), r1.detach()) and torch.equal(d2.cpu(), r2.detach()), "same fp32 arithmetic: bit-exact"
        assert torch.equal(i1.cpu().long(), j1) and torch.equal(i2.cpu().long(), j2)
        w1, w2 = torch.rand(b, n, generator=g), torch.rand(b, m, generator=g)
        ((r1 * w1).sum() + (r2 * w2).sum()).backward()
        ((d1 * w1.to(DEV)).sum() + (d2 * w2.to(DEV)).sum()).backward()
        rel_close(ag.grad, a.grad, 1e-5, 1e-6, "chamfer grad xyz1")
        rel_close(cg.grad, c.grad, 1e-5, 1e-6, "chamfer grad xyz2")
    else:  # full ray-batch size: properties instead of the O(n m) oracle
        sel = torch.arange(0, n, 97)
        dd = ((ag.detach()[0, sel, None, :] - cg.detach()[0, None, :, :]) ** 2)
        ref = (dd[..., 0] + dd[..., 1] + dd[..., 2]).min(1)
        assert torch.equal(d1[0, sel], ref.values) and torch.equal(i1[0, sel].long(), ref.indices)
        s1, s2, k1, _ = chamfer_3DDist()(ag, ag)
        assert float(s1.abs().max()) == 0.0 and torch.equal(k1[0].long().cpu(), torch.arange(n))


def test_attr_mlp_gathered_equals_materialised():
    """l4d_attr_mlp_fwd / _bwd (input rows assembled in the kernel from the direction encoding, the sigma network's output and
    ones) against l4d_attr_gather + l4d_mlp_fwd / _bwd on the materialised [rows, 96] matrix: same MFMAs on the same
    operands (the kernel takes the 16 columns behind the direction encoding in the order of the sigma network's row and
    permutes the weight columns to match: each product is the same, the fp32 accumulation order inside a k-chunk changes) ->
    outputs / activations / gradients equal up to fp16 rounding of reordered fp32 sums; dW up to atomics order."""
    from lidar4d_amd import ops
    n_rays, T, n_geo, in_pad = 37, 64, 15, 96
    P = n_rays * T
    dirs = torch.nn.functional.normalize(det_uniform((n_rays, 3), "gad", -1, 1), dim=-1).to(DEV)
    denc = ops.freq_fwd(((dirs + 1) / 2).contiguous(), 12)
    h = det_uniform((P, 16), "gah", -1, 1).half().to(DEV)
    keep = det_uniform((P,), "gak", 0, 1) > 0.6
    idx_host = torch.nonzero(keep).reshape(-1).to(torch.int32)
    idx_host = idx_host[torch.randperm(idx_host.numel(), generator=torch.Generator().manual_seed(1))]  # arrival order is arbitrary
    M = idx_host.numel()
    idx = torch.zeros(P, dtype=torch.int32, device=DEV)
    idx[:M] = idx_host.to(DEV)
    count = torch.tensor([M], dtype=torch.int32, device=DEV)
    for n_hidden in (1, 2):
        w = det_uniform((64 * in_pad + (n_hidden - 1) * 64 * 64 + 16 * 64,), f"gaw{n_hidden}", -0.3, 0.3).half().to(DEV)
        assert ops.attr_mlp_supported(in_pad, denc.shape[1], n_geo)
        xa = ops.attr_gather(idx, count, P, T, denc, h, n_geo, in_pad)
        y0, act0 = ops.mlp_fwd(xa, w, n_hidden, save_act=True, n_rows=count)
        xr = torch.zeros(P, in_pad, dtype=torch.float16, device=DEV)
        y1, act1 = ops.attr_mlp_fwd(idx, count, P, T, denc, h, n_geo, in_pad, w, n_hidden, save_act=True, x_rows_out=xr)
        y2, _ = ops.attr_mlp_fwd(idx, count, P, T, denc, h, n_geo, in_pad, w, n_hidden, save_act=False)
        assert torch.equal(y1[:M], y2[:M])
        # the stored rows: the materialised matrix with the 16 columns behind the direction encoding in the order [1, g0 .. g14]
        assert torch.equal(xr[:M, :72], xa[:M, :72]) and torch.equal(xr[:M, 73:88], xa[:M, 72:87]) and bool((xr[:M, 72] == 1).all())
        assert torch.equal(xr[:M, 88:], xa[:M, 88:])
        def near(a, b, what):  # equal up to an fp16 ulp on a small fraction of the elements
            a, b = a.float(), b.float()
            d = (a - b).abs()
            assert float(d.max()) <= 2 ** -9 * max(float(b.abs().max()), 1e-6) and float((d > 0).float().mean()) < 0.05, (what, float(d.max()), float((d > 0).float().mean()))
        near(y0[:M], y1[:M], "y"), near(act0[:, :M], act1[:, :M], "act")
        # sigmoid epilogue == l4d_attr_scatter on the stored outputs, bit for bit (same fp16 output, same rounding points)
        a0, c0 = torch.zeros(P, 2, device=DEV), torch.zeros(P, 2, device=DEV)
        ops.attr_scatter(idx, count, P, y1, y1, a0, c0)
        a1, c1 = torch.zeros(P, 2, device=DEV), torch.zeros(P, 2, device=DEV)
        for ch in (0, 1):
            ye, _ = ops.attr_mlp_fwd(idx, count, P, T, denc, h, n_geo, in_pad, w, n_hidden, save_act=False,
                                     attr_dense=a1, attr_compact=c1, channel=ch)
            assert ye is None
        assert torch.equal(a0, a1) and torch.equal(c0[:M], c1[:M]) and float(a1.abs().max()) > 0
        dy = torch.zeros(P, 16, dtype=torch.float16, device=DEV)
        dy[:, 0] = det_uniform((P,), "gady", -1, 1).half().to(DEV)
        g0, g1 = torch.zeros(w.numel(), device=DEV), torch.zeros(w.numel(), device=DEV)
        dx0 = ops.mlp_bwd(xa, act0, dy, w, n_hidden, g0, 1.0 / 128, n_rows=count)
        dx1 = ops.attr_mlp_bwd(xr, count, denc.shape[1], n_geo, act1, dy, w, n_hidden, g1, 1.0 / 128)
        # dx_tail's physical columns: [64 .. 71 | junk (d/d ones), g0 .. g14 | 88 .. 95]
        assert dx1.shape == (P, 32)
        # rows assembled in the backward as well: the same operands as the stored rows -> same dx; dW up to atomics order
        g2 = torch.zeros(w.numel(), device=DEV)
        dx2 = ops.attr_mlp_bwd_gathered(idx, count, P, T, denc, h, n_geo, in_pad, act1, dy, w, n_hidden, g2, 1.0 / 128)
        assert torch.equal(dx1[:M], dx2[:M])
        assert float((g1 - g2).abs().max()) <= 1e-5 * float(g1.abs().max())
        # ... and with the sigmoid-scatter adjoint in front / the sum + scatter into dh behind it inside the kernel:
        # == l4d_attr_scatter_bwd -> two backward launches -> l4d_attr_gather_bwd, bit for bit
        d_attr = det_uniform((P, 2), "gada", -1, 1).to(DEV)
        dyR, dyI = torch.empty(P, 16, dtype=torch.float16, device=DEV), torch.empty(P, 16, dtype=torch.float16, device=DEV)
        ops.attr_scatter_bwd(idx, count, P, d_attr, c1, 128.0, dyR, dyI)
        gR, gI = torch.zeros(w.numel(), device=DEV), torch.zeros(w.numel(), device=DEV)
        dR = ops.attr_mlp_bwd_gathered(idx, count, P, T, denc, h, n_geo, in_pad, act1, dyR, w, n_hidden, gR, 1.0 / 128)
        dI = ops.attr_mlp_bwd_gathered(idx, count, P, T, denc, h, n_geo, in_pad, act1, dyI, w, n_hidden, gI, 1.0 / 128)
        dh_ref = torch.zeros(P, 16, dtype=torch.float16, device=DEV)
        ops.attr_gather_bwd(idx, count, P, dR, dI, in_pad - 64, denc.shape[1] - 64, n_geo, dh_ref, h_layout=True)
        dh_epi = torch.zeros(P, 16, dtype=torch.float16, device=DEV)
        gR2, gI2 = torch.zeros(w.numel(), device=DEV), torch.zeros(w.numel(), device=DEV)
        for ch, gw in ((0, gR2), (1, gI2)):
            assert ops.attr_mlp_bwd_gathered(idx, count, P, T, denc, h, n_geo, in_pad, act1, None, w, n_hidden, gw, 1.0 / 128,
                                             d_attr=d_attr, attr_compact=c1, channel=ch, loss_scale=128.0, dh16=dh_epi,
                                             accumulate=ch == 1) is None
        assert torch.equal(dh_epi, dh_ref) and float(dh_ref.abs().max()) > 0
        assert float((gR - gR2).abs().max()) <= 1e-5 * float(gR.abs().max()) and float((gI - gI2).abs().max()) <= 1e-5 * float(gI.abs().max())
        near(dx0[:M, 64:72], dx1[:M, :8], "dx tile head"), near(dx0[:M, 72:87], dx1[:M, 9:24], "d geo")
        assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max()) and float(g0.abs().max()) > 0
        dh0, dh1 = torch.zeros(P, 16, dtype=torch.float16, device=DEV), torch.zeros(P, 16, dtype=torch.float16, device=DEV)
        ops.attr_gather_bwd(idx, count, P, dx0, dx0, in_pad, denc.shape[1], n_geo, dh0)
        ops.attr_gather_bwd(idx, count, P, dx1, dx1, in_pad - 64, denc.shape[1] - 64, n_geo, dh1, h_layout=True)
        near(dh0, dh1, "dh")
        assert float(dh0.abs().max()) > 0 and bool((dh1[:, 0] == 0).all())
    # empty work list
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    y, _ = ops.attr_mlp_fwd(idx, zero, P, T, denc, h, n_geo, in_pad, w, 2, save_act=False)
    assert y.shape == (P, 16)


def test_mlp_fwd_sigma_epilogue():
    """l4d_mlp_fwd_sigma == l4d_mlp_fwd followed by l4d_sigma_from_h, bit for bit, on every width the model configs use."""
    from lidar4d_amd import ops
    for in_pad, n_hidden, P in ((128, 1, 4099), (176, 2, 515), (64, 1, 33), (32, 2, 1000)):
        x = det_uniform((P, in_pad), f"sx{in_pad}", -1, 1).half().to(DEV)
        w = det_uniform((64 * in_pad + (n_hidden - 1) * 64 * 64 + 16 * 64,), f"sw{in_pad}", -0.3, 0.3).half().to(DEV)
        y0, act0 = ops.mlp_fwd(x, w, n_hidden, save_act=True)
        y1, act1, sigma = ops.mlp_fwd_sigma(x, w, n_hidden, save_act=True)
        assert torch.equal(y0, y1) and torch.equal(act0, act1)
        assert torch.equal(sigma, ops.sigma_from_h(y0)) and float(sigma.min()) > 0
