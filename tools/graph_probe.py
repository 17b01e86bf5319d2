"""Probe (GPU box): does the whole training step capture into a hipGraph under the current environment, and what does a replay
cost against eager launches?   usage: python tools/graph_probe.py [n_rays]   (env: L4D_GRAPH_BATCH, L4D_GRAPH_STREAMS, L4D_STREAMS)"""
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
tr = Trainer(model, data, chamfer=True, flow=True, ema_decay=None)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    tr.train_step(data.batch_for(20))
eager = timed(lambda: tr.train_step(data.batch_for(20)), 10)
print("eager %.3f ms/step" % eager, flush=True)
for f in (20, 21):
    tr.train_step_graphed(f)
    print("captured frame", f, flush=True)
replay = timed(lambda: tr.train_step_graphed(20), 10)
losses = [float(tr.train_step_graphed(21)) for _ in range(3)]
print("PROBE_OK rays %d eager %.3f graph %.3f ms/step losses %s env %s" % (
    n_rays, eager, replay, losses, {k: v for k, v in os.environ.items() if k.startswith("L4D_")}), flush=True)
