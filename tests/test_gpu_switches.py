"""Run-time A/B switches of the round-5 structural changes ("Round 5 added"): every non-default setting is a
code path the default GPU suite never takes.  Each one renders and back-propagates the same batch in a fresh process (the library
reads its switches once) and must reproduce the default build's outputs -- bit for bit in the forward pass (the variants differ in
WHERE a lookup runs and in which order table lines are fetched, not in arithmetic), to fp32 summation order in the gradients."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, json, sys
import torch
sys.path.insert(0, %(root)r)
from lidar4d_amd import LiDAR4D
from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
from lidar4d_amd.trainer import lidar_loss
torch.manual_seed(0)
dev = "cuda"
model = LiDAR4D(near_lidar=KITTI360_SCALE, far_lidar=81 * KITTI360_SCALE).to(dev)
g = torch.Generator(device=dev).manual_seed(11)
with torch.no_grad():  # visible densities and a flow that leaves the current cell
    model.hash_encoder.hash_static.params.copy_((torch.rand(model.hash_encoder.hash_static.params.shape, device=dev, generator=g) - 0.5))
    for hd in model.hash_encoder.hash_dynamic:
        for enc in hd.hash_t:
            enc.params.copy_((torch.rand(enc.params.shape, device=dev, generator=g) - 0.5))
    model.flow_net.grid_enc.params.copy_((torch.rand(model.flow_net.grid_enc.params.shape, device=dev, generator=g) - 0.5) * 2)
data = SyntheticKitti360(dev, num_rays=512, num_frames=51, seed=5)
b = data.batch_for(20)
noise = torch.rand(512, 768, device=dev, generator=g)
out = model.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, num_steps=768, perturb=True, noise=noise)
lidar_loss(out, b["images_lidar"]).backward()
torch.cuda.synchronize()
h = lambda t: hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()
gr = model._store.flat_grad
print(json.dumps({"depth": h(out["depth_lidar"]), "image": h(out["image_lidar"]), "weights": h(out["weights"]),
                  "grad_abs_sum": float(gr.abs().double().sum()), "grad_max": float(gr.abs().max()),
                  "grad_proj": float((gr.double() * torch.linspace(-1, 1, gr.numel(), device=dev, dtype=torch.float64)).sum()),
                  "finite": bool(torch.isfinite(gr).all())}))
"""


def _run(env_extra, want_stderr=False):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return (d, r.stderr) if want_stderr else d


@pytest.fixture(scope="module")
def default_digest():
    return _run({})


@pytest.mark.parametrize("switch", ["L4D_ENC_HS_SPLIT=0", "L4D_HS_PAIRLD=0", "L4D_HG_ORDER=0", "L4D_FLOW_LEVELS=0", "L4D_ENC_SIGMA=0",
                                    "L4D_ENC_PERSISTENT=0", "L4D_MLP_RECOMP_SIGMA=0", "L4D_DH_PARITY=0", "L4D_ENC_HS_SPLIT=0,L4D_HS_PAIRLD=0"])
def test_switch_reproduces_the_default_path(switch, default_digest):
    got = _run(dict(kv.split("=") for kv in switch.split(",")))
    assert got["finite"] and default_digest["finite"]
    for k in ("depth", "image", "weights"):
        assert got[k] == default_digest[k], (switch, k)  # forward: bit-identical
    for k in ("grad_abs_sum", "grad_max", "grad_proj"):  # backward: the same contributions; float atomics order the dW / plane flushes
        a, b = got[k], default_digest[k]
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-12) + (1e-6 * default_digest["grad_abs_sum"] if k == "grad_proj" else 0.0), (switch, k, a, b)



def test_trace_switch_names_every_launch_and_changes_nothing(default_digest):
    """L4D_TRACE=1 (csrc/capi.cpp): every launch of the library is announced on stderr and followed by a stream synchronisation
    whose status is printed -- the tool that finds the kernel behind an asynchronous fault.  Same outputs, same gradients."""
    got, err = _run({"L4D_TRACE": "1"}, want_stderr=True)
    for k in ("depth", "image", "weights"):
        assert got[k] == default_digest[k], k
    for name in ("density_encode_fwd_kernel", "composite_fwd_kernel", "bin_pass1_kernel", "bin_reduce_kernel", "planes_dyn_lds_kernel"):
        assert "[l4d] launch (" + name in err or "[l4d] launch " + name in err, name
    done = [l for l in err.splitlines() if l.startswith("[l4d] done")]
    assert done and all(l.endswith(": ok") for l in done)
