"""bench.py -- training rays/s of the LiDAR4D ray-rendering hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c2-like ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + backward + Adam of ``render(staged=False, perturb=True)`` with the reference's three primary
losses (runner.py:179-213) on one batch of synthetic KITTI-360-shaped rays (64 x 1024 panorama, T = 768 samples per
ray).  Default workload = BASELINE.json configs[2] ("C3": full 4D field -- hash + hex-planes + flow --
16,384 rays/batch/GPU), whose 8-GPU weak-scaled form is configs[3] (131,072 rays/step), the configuration the
headline ">= 10 M training rays/s on 8 x MI355X" is quoted on.  Rank r draws its own ray indices (all ranks step through the same
frame sequence, so that the per-step work is the same everywhere); the flat gradient buffer is SUM-all-reduced over RCCL
once per step.  Inputs are resident in HBM before the timed region.

The JSON line also carries `roofline` (dominant kernel: algorithmic bytes / measured launch time vs the 8 TB/s HBM
peak; per-kernel times come from HIP events recorded around every launch on the launch stream in a separate
profiling pass after the timed region) and, on rank 0 at N = 1, `cpu_baseline` (the oracle, i.e. a port, timed on
the host cores over a bounded ray sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (model kwargs, rays per GPU per step, description)
    "c3": (dict(), 16384, "C3: KITTI-360 seq-4950-shaped full 4D (L=8 hash + hex-planes + flow_field), 16384 rays/batch/GPU training"),
    "c2": (dict(n_levels_hash=16, num_layers_sigma=3), 4096,
           "C2-shaped: L=16 hash + 3-layer-64 sigma MLP, 4096 rays/batch (the reference has no static-only switch: the full 4D field runs)"),
    "c3-4k": (dict(), 4096, "full 4D, 4096 rays/batch/GPU training (staged-chunk size)"),
    "c3-l2": (dict(log2_hashmap_size=15), 16384, "analysis only: C3 work with hash tables shrunk 16x (all tables L2-resident): isolates the cost of L2 misses"),
    "c3-1k": (dict(), 1024, "full 4D, 1024 rays/batch/GPU training (the reference's own num_rays_lidar)"),
    # inference (BASELINE configs[4]): one step = one 64 x 2048 novel-view frame per GPU under no_grad --
    # render(staged=True, 32 chunks of 4096 rays) + U-Net ray-drop refinement + masking (runner.py:438-470) +
    # range image -> points -> chamfer distance / F-score against the ground-truth frame (utils/metrics.py:249-270)
    "c5": (dict(), 64 * 2048, "C5: full LiDAR4D inference, 2048x64 novel-view render (staged) + U-Net ray-drop refinement + chamfer3D eval, one frame/GPU/step"),
}
INFERENCE = {"c5"}


def algorithmic_bytes_per_sample(model):
    """SURVEY.md 8(d) per-sample table-entry traffic of the fused kernels at this model's configuration.
    fwd: fp16 hash entries (8 B at F=4), fp32 channel-last plane texels (32 B);
    bwd: plane values are re-read for the product rule, gradients are fp32 read-modify-write (x2)."""
    he, pe = model.hash_encoder, model.planes_encoder
    L = he.hash_static.meta.n_levels
    nS = pe.layout.n_scales
    static = L * 8 * 8
    dyn_eval = 3 * 2 * L * 4 * 8
    planes_full = 6 * nS * 4 * 32
    planes_dyn = 3 * nS * 4 * 32
    flow = model.flow_net.n_levels * 8 * 16
    fwd = static + 3 * dyn_eval + planes_full + 2 * planes_dyn
    bwd = (planes_full + 2 * planes_dyn) + 2 * ((L * 8 * 16) + (3 * 2 * L * 4 * 16) + (planes_full + 2 * planes_dyn))
    return {"l4d_density_encode_fwd": fwd, "l4d_density_encode_bwd": bwd, "l4d_hashgrid_t_fwd": flow,
            "l4d_hashgrid_t_bwd": 2 * model.flow_net.n_levels * 8 * 32, "hash_static_fwd_only": static}


CPU_BASELINE_CODE = r"""
import json, os, sys, time
import torch
sys.path.insert(0, {root!r})
from oracle import fields_ref, tcnn_ref
tcnn_ref.set_precision("tcnn")
cores = min(os.cpu_count() or 1, 32)          # oversubscribing a big host with tiny torch ops only thrashes
torch.set_num_threads(cores)
scale, num_frames, budget = {scale!r}, {num_frames!r}, {budget!r}
m = fields_ref.LiDAR4D(near_lidar=1.0 * scale, far_lidar=81.0 * scale, num_frames=num_frames)
g = torch.Generator().manual_seed(0)
rd = torch.nn.functional.normalize(torch.randn(1, 1024, 3, generator=g), dim=-1)
ro = torch.zeros(1, 1024, 3)
t = torch.tensor([[0.5]])
def step(n):
    t0 = time.time()
    out = m.render(ro[:, :n], rd[:, :n], t, num_steps=768, perturb=True)
    (out["depth_lidar"].sum() + out["image_lidar"].sum()).backward()
    return time.time() - t0
step(4)                                        # warm-up (allocator, thread pool)
probe = step(8)
n = int(max(8, min(1024, budget / max(probe / 8, 1e-6))))
dt = step(n)
print(json.dumps(dict(value=n / dt, unit="rays/s", cores=cores, kind="port",
      sample="oracle (torch CPU restatement of the reference path, tiny-cuda-nn rounding points) forward+backward of "
             "render() on %d rays x 768 samples, full 4D default config, no optimizer step; %.1f s on %d threads" % (n, dt, cores))))
"""


def cpu_baseline(num_frames, scale, budget_s=20.0, timeout_s=240):
    """Oracle (CPU restatement = a port of the reference path) fwd+bwd on the host cores, in a subprocess with a hard
    timeout so the bench line is always produced."""
    import subprocess
    code = CPU_BASELINE_CODE.format(root=ROOT, scale=scale, num_frames=num_frames, budget=budget_s)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "rays/s", "cores": None, "kind": "port", "sample": "cpu baseline failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "rays/s", "cores": None, "kind": "port", "sample": f"cpu baseline timed out after {timeout_s} s"}


class _QuietStdout:
    """Everything written to file descriptor 1 inside the block goes to stderr instead -- including what native libraries
    print on their own (RCCL announces itself on stdout when the process group comes up) -- so that the JSON line stays
    the only thing this script ever writes to stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    with _QuietStdout():
        line = _run()
    if line is not None:
        print(json.dumps(line), flush=True)


def _run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chamfer", action="store_true", help="add the reference's ray chamfer loss term (runner.py:215-220) to the step")
    ap.add_argument("--flow", action="store_true", help="add the reference's scene-flow consistency loss (runner.py:222-253, opt.flow_loss)")
    ap.add_argument("--sort-rays", action="store_true", help="serve the random pixels of a batch in 8x8-pixel-block order (locality experiment; measured slower)")
    ap.add_argument("--urf", action="store_true", help="add the line-of-sight loss (runner.py:255-276, opt.urf_loss)")
    ap.add_argument("--no-ema", action="store_true", help="skip the parameter EMA update the reference does after every step (--ema_decay 0.95 by default there)")
    ap.add_argument("--profile-steps", type=int, default=2)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # L4D_FORCE_DIST=1: take the multi-rank code path (RCCL init, barrier, gradient all-reduce) even with one rank --
    # lets the data-parallel path be exercised on a single-GPU box (python -m torch.distributed.run --nproc-per-node 1)
    force_dist = os.environ.get("L4D_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from lidar4d_amd import LiDAR4D, _lib
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import Trainer

    model_kw, n_rays, desc = WORKLOADS[args.workload]
    torch.manual_seed(0)  # identical initial replicas on every rank
    model = LiDAR4D(near_lidar=1.0 * KITTI360_SCALE, far_lidar=81.0 * KITTI360_SCALE, num_frames=51, **model_kw).to(dev)
    inference = args.workload in INFERENCE
    data = SyntheticKitti360(dev, W=2048 if inference else 1024, num_rays=n_rays, seed=1000 + rank, sort_pixels=args.sort_rays,
                             frame_seed=1000)  # every rank: its own rays, the same frame sequence (equal work per step)
    trainer = Trainer(model, data, chamfer=args.chamfer, flow=args.flow, urf=args.urf, ema_decay=None if args.no_ema else 0.95)
    if inference:
        from lidar4d_amd.data import KITTI360_FOV
        from lidar4d_amd.metrics import PointsMeter
        model.eval()
        meter = PointsMeter(scale=KITTI360_SCALE, intrinsics=KITTI360_FOV)
        frames = [data.frame((rank + world * k) % data.num_frames) for k in range(4)]  # resident before the timed region
        counter = [0]

        def step():
            fr = frames[counter[0] % len(frames)]
            counter[0] += 1
            _, _, pred_depth = trainer.test_step(fr, refine=True)
            gt = fr["images_lidar"]
            meter.update(pred_depth, gt[..., 2] * gt[..., 0])
    else:
        step = trainer.train_step

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    # ---- per-kernel timing pass (HIP events on the launch stream), outside the timed region ----
    roofline = None
    kernels = {}
    if args.profile_steps > 0:
        # every rank runs the extra steps (they contain the gradient all-reduce); only rank 0 records events
        if rank == 0:
            _lib.PROFILE = []
        for _ in range(args.profile_steps):
            step()
        barrier()
    if rank == 0 and args.profile_steps > 0:
        for name, s, e in _lib.PROFILE:
            k = kernels.setdefault(name, [0, 0.0])
            k[0] += 1
            k[1] += s.elapsed_time(e)
        _lib.PROFILE = None
        per_step = {k: v[1] / args.profile_steps for k, v in kernels.items()}
        # entry points that launch several kernels (the adjoints) are aggregates; the roofline is quoted for the
        # dominant SINGLE kernel, whose name rocprofv3 reports the same way (profiles/)
        aggregates = {"l4d_density_encode_bwd", "l4d_hashgrid_t_bwd", "l4d_planes_relayout"}
        dominant = max((k for k in per_step if k not in aggregates), key=per_step.get)
        launches = kernels[dominant][0] / args.profile_steps
        avg_ms = kernels[dominant][1] / kernels[dominant][0]
        P = min(n_rays, 4096) * 768 if inference else n_rays * 768  # staged inference launches per 4096-ray chunk
        bps = algorithmic_bytes_per_sample(model)
        alg = bps.get(dominant)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and args.workload == "c3":  # PMC passes were collected on the default workload
            traffic = json.load(open(tpath)).get(dominant)
        if alg is not None:
            achieved = alg * P / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "algorithmic_bytes_per_launch": alg * P, "avg_launch_ms": round(avg_ms, 4),
                        "launches_per_step": launches,
                        "rocprof_kernels": "dynhash_fwd_lds_kernel + density_encode_fwd_kernel<true>" if dominant == "l4d_density_encode_fwd" else dominant,
                        "kernel_ms_per_step": {k: round(v, 3) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}}
        else:
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": traffic,
                        "kernel_ms_per_step": {k: round(v, 3) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])}}

    if rank == 0:
        total_rays = n_rays * world * args.steps
        line = {
            "metric": "inference rays/sec (64x2048 novel-view frame)" if inference else "training rays/sec (64x1024 LiDAR panorama)",
            "value": total_rays / dt,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 tables/MFMA operands, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": desc, "rays_per_gpu_per_step": n_rays, "samples_per_ray": 768,
                       "global_rays_per_step": n_rays * world,
                       "parallelism": f"frame-sharded x{world}, no collective" if inference else f"ray-sharded dp{world}, one RCCL gradient all-reduce per step in two phases, overlapped with the tail of the backward pass",
                       "step": "no_grad render(staged=True, max_ray_batch=4096) + U-Net + pano_to_lidar + chamfer/F-score" if inference else
                               "forward + backward + Adam" + ("" if args.no_ema else " + parameter EMA") + ", losses L1 depth + MSE raydrop + MSE intensity" + (" + ray chamfer" if args.chamfer else "") + (" + scene-flow consistency" if args.flow else "") + (" + line-of-sight" if args.urf else "") + ("" if args.chamfer or args.flow or args.urf else " (no chamfer/flow loss)")},
            "roofline": roofline,
        }
        if inference:
            line["eval"] = {"chamfer_distance_m2, f_score@0.05": [float(v) for v in meter.measure()], "frames": meter.N,
                            "note": "random-init field vs synthetic ground truth: exercises the eval path, not a quality claim"}
        if world == 1 and not args.no_cpu_baseline and not inference:
            line["cpu_baseline"] = cpu_baseline(51, KITTI360_SCALE)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    return line if rank == 0 else None


if __name__ == "__main__":
    main()
