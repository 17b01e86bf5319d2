"""Phase clocks of the time-plane adjoint (planes_dyn_lds_kernel) inside real training steps: a library built with -DFB_PHASE_CLOCK
(tools/build_abl.sh fbclk "-DFB_PHASE_CLOCK" field_bwd.hip), a few C3 steps, the per-part share of a sampled wavefront's cycles.
    L4D_LIB=tools/abl/lib_fbclk.so python tools/phase_probe.py [steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar4d_amd import LiDAR4D, _lib  # noqa: E402
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360  # noqa: E402
from lidar4d_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda"
torch.manual_seed(0)
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE, num_frames=51).to(dev)
data = SyntheticKitti360(dev, W=1024, num_rays=16384, seed=1000, frame_seed=1000)
tr = Trainer(model, data, chamfer=True, flow=True, ema_decay=None, init_scale=2048.0)
for _ in range(3):
    tr.train_step()
torch.cuda.synchronize()
lib = _lib.lib()
if not hasattr(lib, "l4d_debug_fb_phase_clk"):
    sys.exit("library without -DFB_PHASE_CLOCK")
lib.l4d_debug_fb_phase_clk(None, 1, None)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(steps):
    tr.train_step()
e.record()
torch.cuda.synchronize()
print(f"{s.elapsed_time(e) / steps:.3f} ms per step")
out = torch.zeros(16, dtype=torch.int64, device=dev)
lib.l4d_debug_fb_phase_clk(C.c_void_p(out.data_ptr()), 1, None)
torch.cuda.synchronize()
a = out.cpu().numpy().astype(np.float64)
n_w = a[15]
names = ["loop tail / head", "coordinates + flow arrive", "PREP: static planes' factors (taps, gvs stores)", "PREP: dynamic-hash columns", "time planes: gradient piece, frame set-up",
         "time planes: row texels loaded + interpolated", "time planes: product rule, adjoint, scans, LDS atomics", "d(flow) stored"]
tot = a[:8].sum()
print(f"sampled wavefronts: {int(n_w)} ({int(n_w / steps)} per launch), {tot / n_w:.0f} cycles each")
for i, nm in enumerate(names):
    print(f"   {nm:58s} {100 * a[i] / tot:6.2f} %   {a[i] / n_w:10.0f}")
