"""Phase clocks of the time-plane adjoint (planes_dyn_lds_kernel, -DFB_PHASE_CLOCK in field_bwd.hip) or of the fused encode
(density_encode_fwd_kernel, -DENC_PHASE_CLOCK in fused.hip) inside real training steps: a library built with the macro
(tools/build_abl.sh clk "-DFB_PHASE_CLOCK -DENC_PHASE_CLOCK" field_bwd.hip fused.hip), a few C3 steps, the per-part share of a sampled
wavefront's cycles.    L4D_LIB=tools/abl/lib_clk.so python tools/phase_probe.py [steps] [fb|enc]"""
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
which = sys.argv[2] if len(sys.argv) > 2 else "fb"
acc_name = {"fb": "l4d_debug_fb_phase_clk", "enc": "l4d_debug_enc_phase_clk"}[which]
dev = "cuda"
torch.manual_seed(0)
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE, num_frames=51).to(dev)
data = SyntheticKitti360(dev, W=1024, num_rays=16384, seed=1000, frame_seed=1000)
tr = Trainer(model, data, chamfer=True, flow=True, ema_decay=None, init_scale=2048.0)
for _ in range(3):
    tr.train_step()
torch.cuda.synchronize()
lib = _lib.lib()
if not hasattr(lib, acc_name):
    sys.exit("library without the phase-clock macro")
acc = getattr(lib, acc_name)
acc(None, 1, None)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(steps):
    tr.train_step()
e.record()
torch.cuda.synchronize()
print(f"{s.elapsed_time(e) / steps:.3f} ms per step")
out = torch.zeros(16, dtype=torch.int64, device=dev)
acc(C.c_void_p(out.data_ptr()), 1, None)
torch.cuda.synchronize()
a = out.cpu().numpy().astype(np.float64)
n_w = a[15]
names_enc = ["loop tail / head", "coordinates + flow arrive", "hex-planes: time rows of the scales, blend, staging", "static grid's columns staged", "dynamic hash: xz / yz columns (+ the frame set-up)",
             "row copy-out to X", "density network + its stores", "dynamic hash: xy stack (gathers + three frames' evaluation)", "hex-planes: static planes' taps of the scales"]
names = ["loop tail / head", "coordinates + flow arrive", "PREP: static planes' factors (taps, gvs stores)", "PREP: dynamic-hash columns", "time planes: gradient piece, frame set-up",
         "time planes: row texels loaded + interpolated", "time planes: product rule, adjoint, scans, LDS atomics", "d(flow) stored"]
if which == "enc":
    names = names_enc
tot = a[:9].sum() if which == "enc" else a[:8].sum()
print(f"sampled wavefronts: {int(n_w)} ({int(n_w / steps)} per launch), {tot / n_w:.0f} cycles each")
for i, nm in enumerate(names):
    print(f"   {nm:58s} {100 * a[i] / tot:6.2f} %   {a[i] / n_w:10.0f}")
