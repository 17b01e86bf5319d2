"""Diagnostic (GPU box): does a REPLAY of the captured training step produce the gradients the eager step produces from the same
state and the same batch?  Random draws are pinned (static batch, no sample jitter, fixed ground-flow time), the training state is
put back before every step, and the flat gradient arena of each replay is compared tensor by tensor with the eager one.
usage: python tools/graph_diff.py [n_rays] [replays]      (env: L4D_STREAMS, PROBE_FLOW, PROBE_CHAMFER)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["L4D_GRAPH_BATCH"] = "outside"
from lidar4d_amd import LiDAR4D, trainer as trainer_mod  # noqa: E402
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360  # noqa: E402
from lidar4d_amd.params import bump_epoch  # noqa: E402
from lidar4d_amd.trainer import Trainer  # noqa: E402

if os.environ.get("PATCH_ZEROS") == "1":  # hypothesis: torch's zero fills are hipMemsetAsync nodes, and memset nodes misbehave in replays
    _empty, _empty_like = torch.empty, torch.empty_like
    torch.zeros = lambda *a, **kw: _empty(*a, **kw).fill_(0)
    torch.zeros_like = lambda t, **kw: _empty_like(t, **kw).fill_(0)
    torch.Tensor.zero_ = lambda self: self.fill_(0)
    torch.Tensor.new_zeros = lambda self, *a, **kw: self.new_empty(*a, **kw).fill_(0)

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(dev)
data = SyntheticKitti360(dev, W=1024, num_rays=n_rays, seed=1000, frame_seed=1000)
tr = Trainer(model, data, chamfer=os.environ.get("PROBE_CHAMFER", "1") == "1", flow=os.environ.get("PROBE_FLOW", "1") == "1", ema_decay=None,
             init_scale=1024.0)
st, opt = model._store, tr.opt
# pin every random draw of a step
batch = {k: (v.contiguous().clone() if torch.is_tensor(v) else v) for k, v in data.batch_for(20).items()}
data.batch_for = lambda frame: batch
render = model.render
model.render = lambda *a, **kw: render(*a, **{**kw, "perturb": False})
tg = torch.tensor([0.37], device=dev)
fl = trainer_mod.flow_loss
trainer_mod.flow_loss = lambda *a, **kw: fl(*a, **{**kw, "t_ground": tg})
for _ in range(6):
    tr.train_step(batch)
opt.device_schedule()


def snapshot():
    return {"flat": st.flat.detach().clone(), "m": opt.exp_avg.clone(), "v": opt.exp_avg_sq.clone(), "steps": opt.steps.clone(),
            "scaler": tr.scaler.state.clone(), "sched": opt.sched.clone(), "count": opt.step_count}


def restore(s):
    with torch.no_grad():
        st.flat.copy_(s["flat"]), opt.exp_avg.copy_(s["m"]), opt.exp_avg_sq.copy_(s["v"]), opt.steps.copy_(s["steps"])
        tr.scaler.state.copy_(s["scaler"]), opt.sched.copy_(s["sched"])
    opt.step_count = s["count"]
    bump_epoch()
    st.refresh16()


snap = snapshot()
restore(snap)
tr.train_step(batch)
torch.cuda.synchronize()
g_e = st.flat_grad.detach().clone()
p_e = st.flat.detach().clone()
restore(snap)
tr.train_step(batch)
torch.cuda.synchronize()
g_e2 = st.flat_grad.detach().clone()


def compare(tag, g, ref):
    bad, worst = [], []
    for name, p, off, n, gi in st.entries:
        if not n:
            continue
        a, b = g[off:off + n], ref[off:off + n]
        if not bool(torch.isfinite(a).all()):
            bad.append(name)
            continue
        d = float((a.double() - b.double()).abs().max()) / max(float(b.abs().max()), 1e-30)
        if d > 1e-3:
            worst.append((name.replace("hash_encoder.", "he.").replace(".params", ""), "%.2e" % d))
    print("%s: non-finite in %s; relative difference > 1e-3 in %s" % (tag, [b.replace("hash_encoder.", "he.") for b in bad[:8]], worst[:10]), flush=True)
    return bool(bad or worst)


compare("eager vs eager (atomics order only)", g_e2, g_e)
restore(snap)
tr.train_step_graphed(20)  # eager warm-up + capture
n_bad = 0
for k in range(replays):
    restore(snap)
    tr.train_step_graphed(20)
    torch.cuda.synchronize()
    n_bad += compare("replay %2d" % k, st.flat_grad, g_e)
    if k == 0:
        print("  parameters after the replayed step vs after the eager step: max |diff| %.3e" % float((st.flat - p_e).abs().max()), flush=True)
print("GRAPH_DIFF rays %d replays %d differing %d  streams mask %s" % (n_rays, replays, n_bad, os.environ.get("L4D_STREAMS", "0")), flush=True)
