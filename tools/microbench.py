"""Per-component GPU timings on ray-ordered KITTI-360-shaped samples (design input, not the bench)."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar4d_amd import ops, LiDAR4D
from lidar4d_amd.gridmeta import GridMeta
from lidar4d_amd.data import SyntheticKitti360, KITTI360_SCALE

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 768
ds = SyntheticKitti360(dev, num_rays=N, num_frames=3)
b = ds.batch_for(1)
lin = torch.linspace(0, 1, T, device=dev)
noise = torch.rand(N, T, device=dev)
t_dev = torch.tensor([0.5], device=dev)
z, xt = ops.sample_rays_xt(b["rays_o_lidar"][0].contiguous(), b["rays_d_lidar"][0].contiguous(), lin, noise, t_dev,
                           KITTI360_SCALE, 81 * KITTI360_SCALE, 1.0)
P = xt.shape[0]
print("P =", P)

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

def one_level(meta, lvl):
    m = GridMeta.__new__(GridMeta)
    m.__dict__.update(meta.__dict__)
    m.n_levels = 1
    for k in ("scale", "res", "size", "hashed"):
        setattr(m, k, [getattr(meta, k)[lvl]])
    m.offset = [0]
    m.n_entries = m.size[0]
    m.n_output_dims = m.n_features
    return m

for name, D, F, log2T, base, maxres, cols in (("static3d", 3, 4, 19, 512, 32768, (0, 1, 2)), ("dyn_xy", 2, 4, 15, 512, 32768, (0, 1)),
                                            ("dyn_xz", 2, 4, 13, 512, 32768, (0, 2)), ("flow3d", 3, 8, 18, 32, 8192, (0, 1, 2))):
    meta = GridMeta(D, 8, F, log2T, base, np.exp2(np.log2(maxres / base) / 7))
    for lvl in range(8):
        m1 = one_level(meta, lvl)
        table = (torch.rand(m1.n_entries * F, device=dev) - 0.5).half()
        out = torch.empty(P, F, dtype=torch.float16, device=dev)
        tf = timeit(lambda: ops.hashgrid_fwd(m1, xt, cols, table, out))
        dout = torch.randn(P, F, device=dev).half()
        grad = torch.zeros(m1.n_entries * F, device=dev)
        tb = timeit(lambda: ops.hashgrid_bwd(m1, xt, cols, dout, grad, 1.0))
        ncorn = 2 ** D
        print(f"{name} lvl{lvl} res {m1.res[0]:6d} hashed {m1.hashed[0]}: fwd {tf:7.3f} ms ({P*ncorn/tf/1e6:8.1f} G gathers/s)  "
              f"bwd {tb:8.3f} ms ({P*ncorn*F/tb/1e6:8.1f} G atomics/s)")

model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(dev)
pe = model.planes_encoder
arena = pe._arena()
for which, nm in ((1, "static"), (2, "dynamic")):
    tf = timeit(lambda: ops.planes_fwd(pe.layout, arena, xt, which))
    d = torch.randn(P, 32, device=dev)
    g = torch.zeros_like(arena)
    tb = timeit(lambda: ops.planes_bwd(pe.layout, arena, xt, which, d if which == 1 else None, d if which == 2 else None, g, False), n=1)
    print(f"planes {nm}: fwd {tf:.3f} ms  bwd {tb:.3f} ms")
