"""Diagnostic (GPU box): does the frame-50 gradient discrepancy of tests/test_gpu_c3_parity.py follow the frame or the rays?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import fields_ref, tcnn_ref
from oracle.detparams import det_uniform, fill_model
from oracle.make_golden import test_rays as make_rays
from lidar4d_amd import LiDAR4D
tcnn_ref.set_precision("tcnn")
S = 0.010504329815187737
kw = dict(near_lidar=S, far_lidar=81 * S, density_scale=30.0)
ref = fill_model(fields_ref.LiDAR4D(**kw), seed=3)
hip = fill_model(LiDAR4D(**kw), seed=3).to("cuda")
n, T = 64, 768
for frame, seed, gkey in ((50, 67, "c3n"), (50, 17, "c3n"), (0, 67, "c3n"), (25, 67, "c3n"), (49, 67, "c3n"), (50, 67, "alt")):
    ro, rd = make_rays(n, seed)
    noise = det_uniform((n, T), f"c3n{frame}", 0.0, 1.0)
    t = torch.tensor([[frame / 50]])
    ref.zero_grad(); hip.zero_grad()
    o_ref = ref.render(ro, rd, t, staged=False, num_steps=T, perturb=True, noise=noise)
    o = hip.render(ro.cuda(), rd.cuda(), t.cuda(), staged=False, num_steps=T, perturb=True, noise=noise.cuda())
    gd_ = det_uniform((1, n), gkey + "gd", -1, 1); gi_ = det_uniform((1, n, 2), gkey + "gi", -1, 1)
    ((o_ref["depth_lidar"] * gd_).sum() + (o_ref["image_lidar"] * gi_).sum()).backward()
    (((o["depth_lidar"] * gd_.cuda()).sum() + (o["image_lidar"] * gi_.cuda()).sum()) * 8192.0).backward()
    worst = {}
    hp = dict(hip.named_parameters())
    for name, p in ref.named_parameters():
        if p.numel() == 0 or p.grad is None or float(p.grad.abs().max()) == 0 or hp[name].grad is None:
            continue
        a, b = hp[name].grad.double().cpu().reshape(-1) / 8192.0, p.grad.double().reshape(-1)
        key = name.split(".")[0] + ("." + name.split(".")[1] if name.startswith(("hash_encoder", "flow_net")) else "")
        e = float((a - b).abs().max() / b.abs().max())
        worst[key] = max(worst.get(key, 0.0), e)
    w = o_ref["weights"]
    sig_max = float((w.max(-1).values).max())
    print(f"frame {frame} rays-seed {seed} grads {gkey}: " + " ".join(f"{k}={v:.1e}" for k, v in worst.items()) +
          f" | max weight {sig_max:.3f} wsum max {float(o_ref['weights_sum_lidar'].max()):.4f}")
