"""Probe (GPU box): does the whole training step capture into a hipGraph under the current environment, does a replay train
(parameters move, gradients finite, optimiser state advances), and what does a replay cost against eager launches?
usage: python tools/graph_probe.py [n_rays]   (env: L4D_GRAPH_BATCH, L4D_STREAMS)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidar4d_amd import LiDAR4D  # noqa: E402
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360  # noqa: E402
from lidar4d_amd.trainer import Trainer  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(dev)
data = SyntheticKitti360(dev, W=1024, num_rays=n_rays, seed=1000, frame_seed=1000)
tr = Trainer(model, data, chamfer=os.environ.get("PROBE_CHAMFER", "1") == "1", flow=os.environ.get("PROBE_FLOW", "1") == "1", ema_decay=None,
             init_scale=float(os.environ.get("PROBE_SCALE", "65536")))
st = model._store


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def report(tag, before):
    torch.cuda.synchronize()
    g = st.flat_grad
    bad = []
    for name, p, off, n, gi in st.entries:
        if n and not bool(torch.isfinite(g[off:off + n]).all()):
            bad.append(name)
    print("%s: |dparam| max %.3e  grad finite %s absmax %.3e  scaler %s  steps max %d  sched %s  non-finite grads in: %s" % (
        tag, float((st.flat - before).abs().max()), not bad, float(torch.nan_to_num(g[:st.numel], nan=0.0, posinf=0.0, neginf=0.0).abs().max()),
        [round(v, 4) for v in tr.scaler.state.tolist()], int(tr.opt.steps.max()),
        None if tr.opt.sched is None else tr.opt.sched.tolist(), bad[:6]), flush=True)


for _ in range(12):  # let the loss scale settle (it backs off from 65536 x 128 while gradients overflow)
    tr.train_step(data.batch_for(20))
b = st.flat.clone()
tr.train_step(data.batch_for(20))
report("eager step", b)
eager = timed(lambda: tr.train_step(data.batch_for(20)), 10)
print("eager %.3f ms/step" % eager, flush=True)
for f in (20, 21):
    b = st.flat.clone()
    tr.train_step_graphed(f)
    report("capture call frame %d (eager step inside)" % f, b)
for k in range(3):
    b = st.flat.clone()
    loss = tr.train_step_graphed(20)
    report("replay %d frame 20 loss %.4f" % (k, float(loss)), b)
replay = timed(lambda: tr.train_step_graphed(20), 10)
b = st.flat.clone()
loss = tr.train_step_graphed(21)
report("replay frame 21 loss %.4f" % float(loss), b)
print("PROBE_OK rays %d eager %.3f graph %.3f ms/step env %s" % (
    n_rays, eager, replay, {k: v for k, v in os.environ.items() if k.startswith(("L4D_", "PROBE_"))}), flush=True)
