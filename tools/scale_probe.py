"""Diagnostic (GPU box): which gradient tensors go non-finite, at which loss scale and on which frame, over the first steps of a
training run (the GradScaler halves its scale at every such step).   usage: python tools/scale_probe.py [n_rays] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar4d_amd import LiDAR4D  # noqa: E402
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360  # noqa: E402
from lidar4d_amd.trainer import Trainer  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(dev)
data = SyntheticKitti360(dev, W=1024, num_rays=n_rays, seed=1000, frame_seed=1000)
tr = Trainer(model, data, chamfer=os.environ.get("PROBE_CHAMFER", "1") == "1", flow=os.environ.get("PROBE_FLOW", "1") == "1", ema_decay=None)
st = model._store
orig_check = tr.scaler.check
info = {}


def check(g):
    bad, amax = [], {}
    for name, p, off, n, gi in st.entries:
        if not n:
            continue
        seg = g[off:off + n]
        fin = torch.isfinite(seg)
        top = name.split(".")[0] + "." + name.split(".")[1] if name.count(".") else name
        if not bool(fin.all()):
            bad.append(name)
        else:
            amax[top] = max(amax.get(top, 0.0), float(seg.abs().max()))
    info["bad"], info["amax"] = bad, amax
    orig_check(g)


tr.scaler.check = check
for it in range(steps):
    frame = data.next_frame()
    scale = tr.scaler.get_scale()
    loss = tr.train_step(data.batch_for(frame))
    if info["bad"] or it % 20 == 0:
        top = sorted(info["amax"].items(), key=lambda kv: -kv[1])[:4]
        print("step %3d frame %2d scale %-9g loss %.4g  non-finite: %s   largest finite |scaled grad|: %s" % (
            it, frame, scale, float(loss), [b.replace("hash_encoder.", "he.").replace(".params", "") for b in info["bad"][:8]],
            [(k, "%.3g" % v) for k, v in top]), flush=True)
print("final scale", tr.scaler.get_scale(), "applied steps", int(tr.opt.steps.max()), "of", steps)
