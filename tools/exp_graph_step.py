"""Experiment (not part of the product path): how much of a training step is launch overhead?  Captures Trainer.train_step for
one fixed frame / fixed batch into a hipGraph (learning rate baked in) and compares replay with eager steps on the same frame.
usage: python tools/exp_graph_step.py [n_rays] [frame]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from lidar4d_amd import LiDAR4D  # noqa: E402
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360  # noqa: E402
from lidar4d_amd.trainer import Trainer  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(dev)
data = SyntheticKitti360(dev, W=1024, num_rays=n_rays, seed=1000, frame_seed=1000)
trainer = Trainer(model, data, chamfer=True, flow=True, ema_decay=None)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    trainer.train_step(data.batch_for(frame))
eager = timed(lambda: trainer.train_step(data.batch_for(frame)), 20)
static = data.batch_for(frame)
eager_static = timed(lambda: trainer.train_step(static), 20)
print("rays %d frame %d: eager %.2f ms/step (fresh batch), %.2f ms/step (fixed batch)" % (n_rays, frame, eager, eager_static), flush=True)
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        trainer.train_step(static)  # warm-up on the side stream, as torch's capture recipe asks
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        loss = trainer.train_step(static)
    replay = timed(g.replay, 20)
    print("graph replay %.2f ms/step; loss %.6f" % (replay, float(loss)), flush=True)
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600], flush=True)
