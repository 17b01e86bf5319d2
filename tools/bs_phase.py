"""Where pass 1 of the sorted scatter spends its cycles: runs the flow grid's gradient scatter (bin_pass1_kernel<3, 2>) on ray-ordered
samples through a library built with -DBS_PHASE_CLOCK (tools/build_abl.sh bsclk "-DBS_PHASE_CLOCK" binscatter.hip) and prints the
per-phase share of the wavefronts' time.   L4D_LIB=tools/abl/lib_bsclk.so python tools/bs_phase.py [rays]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar4d_amd import _lib, ops  # noqa: E402
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360  # noqa: E402
from lidar4d_amd.gridmeta import GridMeta  # noqa: E402

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = 768
ds = SyntheticKitti360(dev, num_rays=N, num_frames=3)
b = ds.batch_for(1)
lin = torch.linspace(0, 1, T, device=dev)
noise = torch.rand(N, T, device=dev)
t_dev = torch.tensor([0.5], device=dev)
z, xt = ops.sample_rays_xt(b["rays_o_lidar"][0].contiguous(), b["rays_d_lidar"][0].contiguous(), lin, noise, t_dev, KITTI360_SCALE, 81 * KITTI360_SCALE, 1.0)
P = xt.shape[0]
names = ["head", "pairs: cell + hash", "pairs: take gradient (vmcnt)", "records + ranks", "reservation consumed", "barrier 1", "scan | copy-out", "barrier 2", "dense level (atomics)", "tail", "merged runs: values, scans", "reserve + stage", "merged / dense: cell, flags, hashes", "merged / dense: take gradient", "# pair levels", "# merged levels", "# dense levels", "-"]
lib = _lib.lib()
have_clk = hasattr(lib, "l4d_debug_bs_phase_clk")
for label, F, log2T, base, maxres in (("flow grid  <3,2>", 8, 18, 32, 8192),):
    meta = GridMeta(3, 8, F, log2T, base, np.exp2(np.log2(maxres / base) / 7))
    width = meta.n_levels * (F // 4 if F == 8 else F)
    dout = (torch.randn(P, width, device=dev) * 1e-3).half()
    if F == 8:
        grads = [torch.zeros(meta.n_entries * F, device=dev) for _ in range(2)]
        fn = lambda: ops.hashgrid_t_bwd(meta, xt, (0, 1, 2), 2, t_dev, dout, grads, 1.0)
    else:
        grad = torch.zeros(meta.n_entries * F, device=dev)
        fn = lambda: ops.hashgrid_bwd(meta, xt, (0, 1, 2), dout, grad, 1.0)
    fn()
    torch.cuda.synchronize()
    if have_clk:
        lib.l4d_debug_bs_phase_clk(None, 1, None)
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"{label}: P = {P}, {s.elapsed_time(e) / 3:.3f} ms per call (both passes and the expansion)")
    if have_clk:
        out = torch.zeros(36, dtype=torch.int64, device=dev)
        lib.l4d_debug_bs_phase_clk(C.c_void_p(out.data_ptr()), 1, None)
        torch.cuda.synchronize()
        a = out.cpu().numpy().astype(np.float64).reshape(2, 18) / 3
        n_wg = ((P + 511) // 512 + 15) // 16  # sampled workgroups
        for w, nm in ((0, "wavefront 1"), (1, "wavefronts 4, 7 (mean)")):
            tot = a[w, :14].sum()
            n_w = n_wg * (1 if w == 0 else 2)
            per = tot / n_w
            print(f"  {nm}: {per:9.0f} clock ticks per wavefront")
            for i in range(17):
                print(f"     {names[i]:32s} {100 * a[w, i] / tot:6.2f} %   {a[w, i] / n_w:9.0f}")
