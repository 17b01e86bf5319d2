"""Size-independent properties of the HIP path at the benchmark size (BASELINE C3: 16,384 rays x 768 samples), where
the oracle is too slow to run: partition of unity and linearity of the hash lookup, compositing invariants, the
sorted (binned) gradient scatter against the plain atomic scatter, staged == unstaged rendering, determinism of the
fixed-point gradient accumulation.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_RAYS, T = 16384, 768


@pytest.fixture(scope="module")
def big():
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    torch.manual_seed(0)
    model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(DEV)
    with torch.no_grad():  # visible densities so that masks are non-trivial
        model.hash_encoder.hash_static.params.uniform_(-0.5, 0.5)
        for hd in model.hash_encoder.hash_dynamic:
            for enc in hd.hash_t:
                enc.params.uniform_(-0.5, 0.5)
    data = SyntheticKitti360(DEV, num_rays=N_RAYS, num_frames=51)
    return model, data


def test_hash_partition_of_unity_and_linearity():
    from lidar4d_amd import ops
    from lidar4d_amd.gridmeta import GridMeta
    meta = GridMeta(3, 8, 4, 19, 512, np.exp2(np.log2(32768 / 512) / 7))
    P = N_RAYS * T
    x = torch.rand(P, 3, device=DEV)
    ones = torch.ones(meta.n_params, dtype=torch.float16, device=DEV)
    out = ops.hashgrid_fwd(meta, x, (0, 1, 2), ones)
    assert float((out.float() - 1).abs().max()) <= 2 ** -10, "corner weights must sum to 1"
    a = (torch.rand(meta.n_params, device=DEV) - 0.5).half()
    b = (torch.rand(meta.n_params, device=DEV) - 0.5).half()
    oa, ob = ops.hashgrid_fwd(meta, x, (0, 1, 2), a).float(), ops.hashgrid_fwd(meta, x, (0, 1, 2), b).float()
    oab = ops.hashgrid_fwd(meta, x, (0, 1, 2), (a.float() + b.float()).half()).float()
    # linear in the table up to the fp16 rounding of table sum and outputs
    assert float((oab - (oa + ob)).abs().max()) < 4e-3


def test_binned_scatter_equals_atomic_scatter():
    """l4d_hashgrid_t_bwd with and without the sorted-scatter workspace (flow-grid shape, F=8)."""
    import ctypes as C
    from lidar4d_amd import ops, _lib
    from lidar4d_amd.gridmeta import GridMeta
    meta = GridMeta(3, 8, 8, 18, 32, np.exp2(np.log2(8192 / 32) / 7))
    P = 1 << 20
    x = torch.rand(P, 4, device=DEV)
    t = torch.tensor([0.37], device=DEV)
    dout = (torch.randn(P, 16, device=DEV) * 0.1).half()
    g_binned = torch.zeros(meta.n_params, device=DEV)
    ops.hashgrid_t_bwd(meta, x, (0, 1, 2), 1, t, dout, [g_binned], 1.0)  # P*8 >= threshold -> binned
    prev = ops.BINNED_SCATTER_MIN_RECORDS
    ops.BINNED_SCATTER_MIN_RECORDS = 1 << 62
    try:
        g_atomic = torch.zeros(meta.n_params, device=DEV)
        ops.hashgrid_t_bwd(meta, x, (0, 1, 2), 1, t, dout, [g_atomic], 1.0)
    finally:
        ops.BINNED_SCATTER_MIN_RECORDS = prev
    scale = float(g_atomic.abs().max())
    assert scale > 0
    # the binned path rounds each contribution to fp16 (2^-11 relative) before the exact int64 sum
    assert float((g_binned - g_atomic).abs().max()) < 2e-3 * scale
    g2 = torch.zeros(meta.n_params, device=DEV)
    ops.hashgrid_t_bwd(meta, x, (0, 1, 2), 1, t, dout, [g2], 1.0)
    dense_lvl0 = meta.size[0] * 8  # level 0 of the flow grid is dense -> float atomics; the rest is exact integer sums
    assert torch.equal(g2[dense_lvl0:], g_binned[dense_lvl0:]), "sorted scatter must be bit-reproducible"


def test_render_invariants_full_size(big):
    model, data = big
    b = data.batch_for(20)
    with torch.no_grad():
        out = model.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, num_steps=T, perturb=False)
        st = model.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=True, max_ray_batch=4096, num_steps=T,
                          perturb=False)
    w, z = out["weights"], out["z_vals"]
    assert w.shape == (N_RAYS, T) and bool((w >= 0).all()) and bool(torch.isfinite(w).all())
    assert float(out["weights_sum_lidar"].max()) <= 1.0 + 1e-5          # alpha compositing never exceeds opacity 1
    assert bool((z[:, 1:] > z[:, :-1]).all())                            # sortedness of sample positions
    assert float((out["depth_lidar"].view(-1) - (w * z).sum(-1)).abs().max()) < 1e-5
    assert bool((out["image_lidar"] >= 0).all()) and bool((out["image_lidar"] <= 1.0 + 1e-5).all())
    cnt = int(out["mask_count"])
    idx = out["mask_idx"][:cnt].long()
    assert cnt == int((w > 1e-4).sum()) and bool((w.view(-1)[idx] > 1e-4).all())   # compaction == dense mask
    assert idx.unique().numel() == cnt
    # staged (4 chunks of 4096 rays, renderer.py:159-177) == unstaged
    assert torch.equal(st["depth_lidar"], out["depth_lidar"]) and torch.equal(st["image_lidar"], out["image_lidar"])
    # ... and == staged with the chunks replayed from one captured hipGraph (first chunk eager, second captured, the rest
    # replayed); another frame reuses the graph (the time stays on the device); changed parameters drop it
    model.graph_staged = True
    try:
        with torch.no_grad():
            sg = model.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=True, max_ray_batch=4096, num_steps=T, perturb=False)
            assert model._chunk_graph["key"] is not None
            assert torch.equal(sg["depth_lidar"], out["depth_lidar"]) and torch.equal(sg["image_lidar"], out["image_lidar"])
            b2 = data.batch_for(33)
            ref2 = model.render(b2["rays_o_lidar"], b2["rays_d_lidar"], b2["time"], staged=False, num_steps=T, perturb=False)
            sg2 = model.render(b2["rays_o_lidar"], b2["rays_d_lidar"], b2["time"], staged=True, max_ray_batch=4096, num_steps=T, perturb=False)
            assert torch.equal(sg2["depth_lidar"], ref2["depth_lidar"]) and torch.equal(sg2["image_lidar"], ref2["image_lidar"])
            key = model._chunk_graph["key"]
            model.sigma_net.params.mul_(1.0)  # same values, new version: the fp16 copies will be rebuilt -> graph dropped
            sg3 = model.render(b2["rays_o_lidar"], b2["rays_d_lidar"], b2["time"], staged=True, max_ray_batch=4096, num_steps=T, perturb=False)
            assert model._chunk_graph["key"] != key
            assert torch.equal(sg3["depth_lidar"], ref2["depth_lidar"]) and torch.equal(sg3["image_lidar"], ref2["image_lidar"])
    finally:
        model.graph_staged = False


def test_backward_deterministic_and_finite(big):
    model, data = big
    from lidar4d_amd.trainer import lidar_loss
    b = data.batch_for(7)
    noise = torch.rand(N_RAYS, T, device=DEV)
    grads = []
    for _ in range(2):
        model.zero_grad()
        out = model.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, num_steps=T, perturb=True, noise=noise)
        lidar_loss(out, b["images_lidar"]).backward()
        g = model._store.flat_grad
        assert bool(torch.isfinite(g).all())
        grads.append(g.clone())
    st = model._store
    # static hash grid: sorted scatter = exact integer sums with a single owner per table segment -> bit-identical
    off, n = st.by_param[id(model.hash_encoder.hash_static.params)]
    assert torch.equal(grads[0][off:off + n], grads[1][off:off + n])
    # planes / dynamic hash: exact inside a workgroup, but the per-workgroup flushes are float atomics -> ulp-level noise
    scale = float(grads[0].abs().max())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale
    assert float(grads[0].abs().sum()) > 0
