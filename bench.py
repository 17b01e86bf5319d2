"""bench.py -- training rays/s of the LiDAR4D ray-rendering hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c2|c5 ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

``--gpus N`` (N > 1) without a torchrun environment re-launches this script under torch.distributed.run with N ranks on
127.0.0.1 (and fails loudly if the node has fewer than N GPUs); it never reports fewer GPUs than it was asked for.

A step = the reference's training step (model/runner.py:166-253,474-551): ``render(staged=False, perturb=True)`` of one
batch of synthetic KITTI-360-shaped rays (64 x 1024 panorama, T = 768 samples per ray), the three primary losses (L1 depth,
MSE ray-drop, MSE intensity: runner.py:179-213) + the ray-chamfer term (runner.py:215-220, always on in the reference)
+ the scene-flow consistency loss (runner.py:222-253, ``--flow_loss`` defaults to True there), backward, GradScaler
check / skip / update, Adam.  The parameter EMA is updated once per epoch (runner.py:534-535).  ``--no-chamfer --no-flow``
gives the bare three-loss step; the default run also times it and reports it under ``variants``.
Default workload = BASELINE.json configs[2] ("C3": full 4D field, 16,384 rays/batch/GPU), whose 8-GPU weak-scaled form is
configs[3] (131,072 rays/step), the configuration the headline ">= 10 M training rays/s on 8 x MI355X" is quoted on.
Rank r draws its own ray indices (all ranks step through the same frame sequence, so that the per-step work is the same
everywhere); the flat gradient buffer is SUM-all-reduced over RCCL once per step.  Inputs are resident in HBM before the
timed region.

The ONE stdout line stays under 4 KB (compact_line): the contract's fields plus scalar summaries; everything listed below in
full goes to the side file bench_detail.json (L4D_BENCH_DETAIL), whose path is the line's ``detail`` field.

Besides the contract's fields the line / the detail record carry (rank 0, N = 1):
  roofline          the dominant single kernel against ITS roof (see ``kernel_models``): HBM-streaming kernels against the
                    8 TB/s HBM peak with their compulsory bytes; gather kernels against the 34.5 TB/s L2 peak with the
                    table-entry bytes the algorithm touches (the tables, 97 MB, sit in L2 / Infinity Cache, so dividing those
                    bytes by the HBM peak is meaningless) plus their compulsory HBM rate; the LDS kernel against the LDS peak.
                    Every duration is measured live: HIP events around every kernel launch on the launch stream
                    (csrc/common.h L4D_LAUNCH), in a separate pass after the timed region.
  roofline_kernels  the same for every modelled kernel, sorted by time per step
  hash_encoder      the hash lookup alone (north_star: ">= 40 % HBM roofline for the hash encoder"): the static 3-D grid at
                    L = 8 and L = 16 on the batch's ray-ordered samples, 64 B/level/sample of table entries / time / 8 TB/s
  mfma              algorithmic TFLOP/s of the MLP kernels against the 2.5 PFLOP/s dense fp16 peak (+ the PMC busy figure
                    from profiles/ when a rocprofv3 --pmc pass of this build has been committed)
  parity            depth / intensity / ray-drop error and mask-set equality against the oracle on 64 rays (default model)
  variants          the three-loss step and a trained-state measurement (after --trained-steps further steps)
  cpu_baseline      the oracle (a port) timed on the host cores over a bounded ray sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0     # HBM3E 8.0 TB/s spec (6.3 TB/s measured achievable)
L2_PEAK_GBS = 34500.0     # aggregate L2 bandwidth, 8 XCDs
LDS_PEAK_GBS = 150000.0   # ds_read_b64/b128 streaming, 256 CUs
MFMA_F16_PEAK_TFLOPS = 2500.0
GATHER_PEAK_G = 292.0     # measured here (tools/ubench/gather.hip, gather_policy.hip; profiles/r02_ubench_gather.txt, r05_ubench_gather_policy.txt):
                          # G lane-loads/s of a wave-level gather whose 64 lanes touch 64 different L2-resident lines -- 4, 8 or 16 B per lane and
                          # every cache policy alike (= the L2's 34 TB/s in 128-byte lines); 65 G/s when they come from the Infinity Cache

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


def ops_attr_recompute(n_hidden):
    from lidar4d_amd import ops
    return ops.attr_mlp_recompute_supported(n_hidden)


def kernel_models(model, P, M):
    """Per-kernel roof and byte model: {kernel name: dict(bound, bytes, [alg], note)} per LAUNCH at P sample points and M
    attribute rows (samples with weight > 1e-4).  ``bytes`` is what the roof is charged with:
      bound "hbm": COMPULSORY HBM bytes (inputs read once + outputs written once; tables excluded: they are cache resident)
      bound "fabric": gather kernels.  ``bytes`` = table-entry bytes the algorithm gathers (fp16 hash entries 8 / 16 B, fp32
                   plane texels 32 B); ``hbm`` = the kernel's compulsory HBM bytes.  The tables (97 MB) sit in L2 / Infinity Cache, and a
                   gather moves a 128-byte LINE at whichever boundary it crosses, so the roof that binds is the fabric between L2 and
                   Infinity Cache / HBM (8 TB/s, the HBM figure): the headline fraction is COUNTER bytes / time / 8 TB/s when a counter
                   file of this build exists (profiles/hbm_traffic_rNN.json), else the compulsory HBM bytes; the algorithmic bytes
                   against the 34.5 TB/s L2 peak are reported next to it (``algorithmic``)
      bound "lds": table-entry bytes served from LDS
    Kernel names are the launch sites' (csrc), the same rocprofv3's kernel trace shows."""
    he, pe = model.hash_encoder, model.planes_encoder
    L = he.hash_static.meta.n_levels
    nS = pe.layout.n_scales
    in_pad = model.sigma_net.in_pad
    a_pad = model.intensity_net.in_pad
    n_dyn = 3 * L
    Lf = model.flow_net.n_levels
    nh_s, nh_a = model.sigma_net.n_hidden_layers, model.intensity_net.n_hidden_layers
    planes_full, planes_dyn = 6 * nS * 4 * 32, 3 * nS * 4 * 32
    enc_alg = L * 8 * 8 + 3 * (2 * L * 4 * 8) + planes_full + 2 * planes_dyn   # static + xy stack x 3 frames + planes
    lds_alg = 2 * 3 * (2 * L * 4 * 8)                                            # xz, yz stacks x 3 frames
    X = 2 * in_pad
    m = {}
    fl = lambda pad, nh: 2 * (pad * 64 + (nh - 1) * 64 * 64 + 64 * 16)
    # gathers always issued: static grid 8 corners + current-frame xy stack 4 corners per level; the two warped frames' 4 + 4 are
    # issued only where the warped point leaves the current point's cell (all reused while the flow is zero, as at initialisation)
    enc = dict(bound="fabric", bytes=enc_alg * P, hbm=(16 + 32 + X + 2 * 2 * L) * P, gathers=(L * 8 + L * 4) * P,
               note="planes + static hash + xy dynamic hash gathers (time planes via per-call 1-D rows, both time slices of a corner "
                    "in one 16-B load); row staged in LDS, written once")
    m["density_encode_fwd_kernel<true, true, 0>"] = enc
    m["density_encode_fwd_kernel<true, false, 0>"] = enc
    m["density_encode_fwd_kernel<true, true, 0, 0>"] = m["density_encode_fwd_kernel<true, true, 0, 2>"] = enc
    # round 5: the static grid through a level-major pre-pass (hashgrid_fwd_levels_kernel); the encode kernel gathers the hex-planes and
    # the xy stack and takes the static grid's columns from the pre-pass
    m["density_encode_fwd_kernel<true, true, 0, 1>"] = dict(bound="fabric", bytes=(enc_alg - L * 8 * 8) * P, hbm=(16 + 32 + X + 2 * 2 * L + 8 * L) * P, gathers=L * 4 * P,
                                                           note="planes + xy dynamic hash gathers; static-grid columns read from the level-major pre-pass (8 B per level); row staged in LDS, written once")
    m["density_encode_fwd_kernel<true, true, 0, 1, true>"] = dict(
        bound="fabric", bytes=(enc_alg - L * 8 * 8) * P, hbm=(16 + 32 + X + 2 * 2 * L + 8 * L + 32 + 4) * P, gathers=L * 4 * P, flops=fl(in_pad, nh_s) * P,
        note="planes + xy dynamic hash gathers, static-grid columns from the level-major pre-pass, row staged in LDS and written once, AND the density "
             "network's forward pass on the staged rows as epilogue (y + sigma out; the hidden activations are recomputed by the backward)")
    hs_lv = dict(bound="fabric", bytes=L * 8 * 8 * P, hbm=(L * 12 + L * 8) * P, gathers=L * 6 * P,  # (address slots: 4 pair loads + a neighbour load on half the lanes)
                 note="static 3-D grid, one level at a time chip-wide (every L2 holds that level's 4 MB table): 8 corners x 8 B per level; "
                      "x-neighbour pairs in one 16-byte load where aligned (6 address slots per level instead of 8)")
    m["hashgrid_fwd_levels_kernel<3, 4, true>"] = m["hashgrid_fwd_levels_kernel<3, 4>"] = hs_lv
    m["hashgrid_t_fwd_levels_kernel<3, 8>"] = dict(bound="fabric", bytes=Lf * 8 * 16 * P, hbm=(Lf * 12 + Lf * 4) * P, gathers=Lf * 8 * P,
                                                   note="flow grid + interpT, one level at a time chip-wide, level-major fp16 output (rows assembled by hashgrid_rows_from_levels_kernel)")
    m["hashgrid_rows_from_levels_kernel<2>"] = dict(bound="hbm", bytes=(Lf * 4 + Lf * 4) * P, note="level-major columns in, rows out")
    m["density_encode_fwd_kernel<false, false, 0>"] = dict(bound="fabric", bytes=(enc_alg + lds_alg) * P, hbm=(16 + 32 + X) * P, note="all gathers direct")
    m["density_encode_fwd_kernel<false, true, 0>"] = m["density_encode_fwd_kernel<false, false, 0>"]
    m["dynhash_fwd_lds_kernel"] = dict(bound="lds", bytes=lds_alg * P, hbm=(2 * L * 16 + 2 * 2 * L) * P,
                                       note="xz / yz HashGridT stacks from LDS-resident slice tables; one streaming pass of xt / flow per (plane, level)")
    m["hashgrid_t_fwd_kernel<3, 8, true>"] = dict(bound="fabric", bytes=Lf * 8 * 16 * P, hbm=(Lf * 16 + 2 * Lf * 2) * P, gathers=Lf * 8 * P, note="flow grid + interpT")
    it_s, nf = in_pad // 16, model.flow_net.n_hidden
    # sigma network: x + y + saved activations; backward also writes dx
    m[f"mlp_fwd_kernel<{it_s}, {nh_s}>"] = dict(bound="hbm", bytes=(X + 32 + 4 + nh_s * 128) * P, flops=fl(in_pad, nh_s) * P, note="x + y + sigma (exp epilogue) + saved activations")
    m[f"mlp_bwd_kernel<{it_s}, {nh_s}, 0, {it_s}, true>"] = dict(bound="hbm", bytes=(X + nh_s * 128 + 32 + X) * P, flops=2 * fl(in_pad, nh_s) * P,
                                                               note="x + activations + dy read, dx written (algorithmic flops: dX + dW)")
    m[f"mlp_bwd_kernel<{it_s}, {nh_s}, 0, {it_s}, true, true>"] = dict(bound="hbm", bytes=(X + 32 + X) * P, flops=3 * fl(in_pad, nh_s) * P,
                                                                     note="x + dy read, dx written; hidden activations recomputed (algorithmic flops: fwd + dX + dW)")
    # flow network: activations are recomputed in the backward, not stored
    m[f"mlp_fwd_kernel<1, {nf}>"] = dict(bound="hbm", bytes=(32 + 32) * P, flops=fl(16, nf) * P, note="x in, y out (no activations stored)")
    m[f"mlp_bwd_kernel<1, {nf}, 0, 1, true, true>"] = dict(bound="hbm", bytes=(32 + 32 + 32) * P, flops=3 * fl(16, nf) * P,
                                                         note="x + dy read, dx written; activations recomputed (algorithmic flops: fwd + dX + dW)")
    # attribute networks on the work list: rows assembled in the kernel (index + sigma-net output row), geo-feature gradient only
    it_a = a_pad // 16
    m[f"mlp_fwd_kernel<{it_a}, {nh_a}, true, false, 2>"] = dict(bound="hbm", bytes=(4 + 32 + 8 + nh_a * 128) * M, flops=fl(a_pad, nh_a) * M,
                                                                note="idx + h row in (direction encoding per ray: cache resident), sigmoid epilogue (dense + compact value) + activations out")
    m[f"mlp_fwd_kernel<{it_a}, {nh_a}, true, true, 2>"] = dict(bound="hbm", bytes=(4 + 32 + 8 + nh_a * 128 + 2 * a_pad) * M, flops=fl(a_pad, nh_a) * M,
                                                               note="the same + the assembled rows stored once for both networks' backward")
    m[f"mlp_bwd_kernel<{it_a}, {nh_a}, 0, {it_a}, true, false, false, 4>"] = dict(bound="hbm", bytes=(2 * a_pad + nh_a * 128 + 32 + 64) * M,
                                                                                flops=2 * fl(a_pad, nh_a) * M, note="stored rows + activations + dy in, 32 gradient columns out")
    m[f"mlp_bwd_kernel<{it_a}, {nh_a}, 0, {it_a}, true, false, true, 4>"] = dict(bound="hbm", bytes=(4 + 32 + nh_a * 128 + 32 + 64) * M,
                                                                               flops=2 * fl(a_pad, nh_a) * M, note="idx + h row (rows assembled again) + activations + dy in, 32 gradient columns out")
    keep_act = not ops_attr_recompute(nh_a)
    if not keep_act:  # the forward stores no activations (the backward recomputes them)
        m[f"mlp_fwd_kernel<{it_a}, {nh_a}, true, false, 2>"] = dict(bound="hbm", bytes=(4 + 32 + 8) * M, flops=fl(a_pad, nh_a) * M,
                                                                    note="idx + h row in (direction encoding per ray: cache resident), sigmoid epilogue (dense + compact value) out; no activations stored")
    m[f"mlp_bwd_kernel<{it_a}, {nh_a}, 0, {it_a}, true, true, true, 4, true>"] = dict(
        bound="hbm", bytes=(4 + 32 + 8 + 8 + 48) * M, flops=3 * fl(a_pad, nh_a) * M,
        note="idx + h row + the two sigmoid-adjoint factors in, activations recomputed; geo-feature gradient into dh (stored by the first network, read + stored by the second)")
    m[f"mlp_bwd_kernel<{it_a}, {nh_a}, 0, {it_a}, true, false, true, 4, true>"] = dict(
        bound="hbm", bytes=(4 + 32 + 8 + 8 + nh_a * 128 + 48) * M, flops=2 * fl(a_pad, nh_a) * M,
        note="idx + h row + the two sigmoid-adjoint factors + activations in; geo-feature gradient into dh (stored by the first network, read + stored by the second)")
    m[f"mlp_fwd_kernel<{it_a}, {nh_a}>"] = dict(bound="hbm", bytes=(2 * a_pad + 32 + nh_a * 128) * M, flops=fl(a_pad, nh_a) * M, note="materialised input rows")
    m[f"mlp_bwd_kernel<{it_a}, {nh_a}, 0, {it_a}, true>"] = dict(bound="hbm", bytes=(4 * a_pad + nh_a * 128 + 32) * M, flops=2 * fl(a_pad, nh_a) * M, note="materialised input rows")
    rec = 4 * 12  # one 12-byte record per x-neighbour PAIR of corners (upper bound: equal-cell runs along a ray are merged first)
    m["bin_pass1_kernel<3, 4>"] = dict(bound="hbm", bytes=(16 + 8 * L + L * rec) * P, note="static grid: xt + dX columns read, records appended to the bin-major lists (upper bound)")
    m["bin_reduce_kernel<3, 4, DEF>"] = dict(bound="hbm", bytes=L * rec * P, note="static grid: the bin's record lists streamed (whole lines, once), segments reduced in LDS int64 accumulators")
    rec2 = 4 * 8
    # (the flow grid's coarse levels -- 32 ... 345 cells per axis -- hold 5 ... 59 consecutive samples of a ray per cell: their runs
    # are merged before a record is formed, so far fewer than 4 records per sample and level exist: an upper bound)
    m["bin_pass1_kernel<3, 2>"] = dict(bound="hbm", bytes=(16 + 4 * Lf + Lf * rec2) * P, upper_bound=True, note="flow grid records (upper bound: merged runs emit fewer)")
    m["bin_reduce_kernel<3, 2, DEF>"] = dict(bound="hbm", bytes=Lf * rec2 * P, upper_bound=True, note="flow grid records (upper bound: merged runs emit fewer)")
    m["field_bwd_prep_kernel"] = dict(bound="hbm", bytes=(16 + X + 3 * nS * 16 + 2 * n_dyn) * P, note="dX row read, plane factors + transposed dyn gradient written")
    m["planes_dyn_lds_kernel<true, false>"] = m["planes_dyn_lds_kernel<false, false>"] = dict(bound="hbm", bytes=(16 + 32 + X + 32) * P, note="xt + flow + dX rows (128-B lines) read, d(flow) written; LDS int32 accumulation")
    m["planes_dyn_lds_kernel<true, true>"] = dict(bound="hbm", bytes=(16 + 32 + X + 32 + 3 * nS * 16 + 2 * n_dyn + 12) * P,
                                                  note="the same + the preparation pass' outputs (static-plane factors, transposed dyn gradient, SoA coordinates) from the one read of the dX rows")
    m["planes_dyn_lds_wide_kernel"] = m["planes_dyn_lds_kernel<true, true>"]  # (more than 24 dynamic-hash columns: the C2-shaped model)
    m["planes_static_lds_kernel"] = dict(bound="hbm", bytes=(3 * nS * 16 + 3 * nS * 8) * P, note="plane-major factors read once per (scale, plane)")
    m["dynhash_lds_kernel"] = dict(bound="hbm", bytes=(2 * n_dyn + n_dyn * 8) * P, per_step=True,
                                   note="transposed gradient + 2 coordinates per (plane, level) pass; two launches per step (levels that fit a 64 KB window / larger ones), modelled together")
    m["sigma_bwd_rows_kernel"] = dict(bound="hbm", bytes=(4 + 4 + 32) * P, note="sigma, d_sigma in; whole dh rows out")
    m["composite_fwd_kernel"] = dict(bound="hbm", bytes=(4 + 4 + 4 + 4) * P, note="sigma, z in; weights, index out")
    m["composite_bwd_kernel"] = dict(bound="hbm", bytes=(4 * 3 + 8 + 4 + 8) * P, note="")
    m["sample_rays_xt_kernel"] = dict(bound="hbm", bytes=(4 + 4 + 16) * P, note="noise in; z, xt out")
    m["attr_gather_kernel"] = dict(bound="hbm", bytes=(4 + 32 + 2 * a_pad) * M, note="")
    n_par = model._store.numel
    m["adam_ranges_kernel"] = dict(bound="hbm", bytes=30 * n_par, upper_bound=True,
                                   note="p, g, m, v read; p, m, v, fp16 copy written (gated-off time slices are skipped: the model is an upper bound)")
    return m


CPU_BASELINE_CODE = r"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, {root!r})
from oracle import fields_ref, tcnn_ref
from oracle.detparams import det_uniform, fill_model
from oracle.make_golden import test_rays
tcnn_ref.set_precision("tcnn")
host_cores = os.cpu_count() or 1
cores = min(host_cores, 32)                    # threads actually used (oversubscribing a big host with tiny torch ops only thrashes)
torch.set_num_threads(cores)
scale, num_frames, budget, out_path = {scale!r}, {num_frames!r}, {budget!r}, {out_path!r}
g = torch.Generator().manual_seed(0)
rd = torch.nn.functional.normalize(torch.randn(1, 1024, 3, generator=g), dim=-1)
ro = torch.zeros(1, 1024, 3)
t = torch.tensor([[0.5]])
def median3(fn, n):                            # BASELINE.md section 3: 1 warm-up + 3 timed repetitions, median
    fn(n)
    return sorted(fn(n) for _ in range(3))[1]
def fit(fn, share):                            # t(n) = a + b n from two sizes: 8 rays and as many as warm-up + 3 repetitions fit into `share` seconds
    fn(4)                                      # (allocator, thread pool)
    t8, t24 = fn(8), fn(24)
    b = max((t24 - t8) / 16, 1e-6)
    a = max(t8 - 8 * b, 0.0)
    n = int(max(32, min(1024, (share / 4 - a) / b)))
    tn = median3(fn, n)
    b = max((tn - t8) / (n - 8), 1e-9)
    return n, tn, max(t8 - 8 * b, 0.0), b
# (b) training step of the default 4D model (C3 / C4's per-GPU work): forward + backward + Adam, the three primary losses' shape
m = fields_ref.LiDAR4D(near_lidar=1.0 * scale, far_lidar=81.0 * scale, num_frames=num_frames)
opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
def step(n):                                   # forward + backward of n rays (linear in n)
    t0 = time.time()
    opt.zero_grad(set_to_none=True)
    out = m.render(ro[:, :n], rd[:, :n], t, num_steps=768, perturb=True)
    (out["depth_lidar"].sum() + out["image_lidar"].sum()).backward()
    return time.time() - t0
def adam(_):                                   # the optimiser step over all 46.5 M parameters (constant per step)
    t0 = time.time()
    opt.step()
    return time.time() - t0
n, dt, a_fb, b_fb = fit(step, 0.6 * budget)
dt_adam = median3(adam, 0)
step_1024 = a_fb + 1024 * b_fb + dt_adam            # the reference's own batch (num_rays_lidar = 1024): what BASELINE.md section 3 (b) asks for
# (a) BASELINE configs[0] (C1): L = 4 hash grid + 2-layer-64 sigma network, forward only, render(staged=True) on a slice of the
# 64 x 1024 frame, extrapolated linearly to its 65,536 rays (reference path renderer.py:142-186)
m1 = fields_ref.LiDAR4D(near_lidar=1.0 * scale, far_lidar=81.0 * scale, num_frames=num_frames, n_levels_hash=4, num_layers_sigma=2)
def fwd(n1):
    t0 = time.time()
    with torch.no_grad():
        m1.render(ro[:, :n1], rd[:, :n1], t, staged=True, num_steps=768, perturb=False)
    return time.time() - t0
n1, dt1, a_f, b_f = fit(fwd, 0.4 * budget)
res = dict(value=1024 / step_1024, unit="rays/s", cores=cores, host_cores=host_cores, kind="port",
      sample="oracle port (torch CPU, tcnn rounding points): training step fwd + bwd + Adam at 1,024 rays x 768 samples, default 4D config, = "
             "a + 1024 b + Adam, with fwd + bwd time a + b n fitted to 8 rays and a %d-ray sample (%.2f s; a = %.2f s, b = %.4f s/ray) and Adam over all "
             "parameters %.2f s; each 1 warm-up + 3 repetitions, median; %d threads of %d host cores" % (n, dt, a_fb, b_fb, dt_adam, cores, host_cores),
      c1_forward=dict(value=65536 / (a_f + 65536 * b_f), unit="rays/s", frame_s=a_f + 65536 * b_f,
                      sample="C1: L=4 hash + 2-layer sigma net, forward only, render(staged=True), %d-ray slice of the 64x1024 frame, "
                             "median of 3: %.2f s; frame_s = a + 65,536 b with a = %.2f s, b = %.5f s/ray fitted to 8 and %d rays" % (n1, dt1, a_f, b_f, n1)))
# parity reference (checker role): the default model with deterministic, visible parameters, 64 rays x 768 samples, frame 25
m.density_scale = 30.0
fill_model(m, seed=3)
pro, prd = test_rays(64, 42)
noise = det_uniform((64, 768), "benchpar", 0.0, 1.0)
with torch.no_grad():
    o = m.render(pro, prd, torch.tensor([[25 / 50]]), num_steps=768, perturb=True, noise=noise)
np.savez(out_path, depth=o["depth_lidar"].numpy(), image=o["image_lidar"].numpy(), mask=o["mask"].numpy(),
         weights=o["weights"].numpy())
print(json.dumps(res))
"""


def cpu_baseline(num_frames, scale, out_path, budget_s=16.0, timeout_s=300):
    """Oracle (CPU restatement = a port of the reference path) fwd+bwd on the host cores, in a subprocess with a hard
    timeout so the bench line is always produced; the same subprocess writes the parity reference outputs."""
    code = CPU_BASELINE_CODE.format(root=ROOT, scale=scale, num_frames=num_frames, budget=budget_s, out_path=out_path)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "rays/s", "cores": None, "kind": "port", "sample": "cpu baseline failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "rays/s", "cores": None, "kind": "port", "sample": f"cpu baseline timed out after {timeout_s} s"}


def parity_against(ref_path, dev, scale):
    """HIP path vs the oracle outputs written by the cpu_baseline subprocess: same model / fill / rays / noise."""
    import numpy as np
    from lidar4d_amd import LiDAR4D
    from oracle.detparams import det_uniform, fill_model          # deterministic fills + rays only (numpy); the oracle's
    from oracle.make_golden import test_rays                        # model code ran in the cpu_baseline subprocess
    ref = np.load(ref_path)
    m = fill_model(LiDAR4D(near_lidar=1.0 * scale, far_lidar=81.0 * scale, num_frames=51, density_scale=30.0), seed=3).to(dev)
    pro, prd = test_rays(64, 42)
    noise = det_uniform((64, 768), "benchpar", 0.0, 1.0)
    with torch.no_grad():
        o = m.render(pro.to(dev), prd.to(dev), torch.tensor([[25 / 50]], device=dev), num_steps=768, perturb=True, noise=noise.to(dev))
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    depth, image = o["depth_lidar"].cpu().numpy(), o["image_lidar"].cpu().numpy()
    cnt = int(o["mask_count"])
    got = set(o["mask_idx"][:cnt].tolist())
    want = set(np.nonzero(ref["mask"].reshape(-1))[0].tolist())
    w = ref["weights"].reshape(-1)
    off_threshold = [i for i in got ^ want if abs(float(w[i]) - 1e-4) >= 1e-7]
    return {"rays": 64, "samples_per_ray": 768, "vs": "oracle/fields_ref (tiny-cuda-nn rounding points), default C3 model, frame 25",
            "depth_rel_err": rel(depth, ref["depth"]), "raydrop_rel_err": rel(image[..., 0], ref["image"][..., 0]),
            "intensity_rel_err": rel(image[..., 1], ref["image"][..., 1]),
            "depth_rmse_vs_ref": float(np.sqrt(np.mean((depth - ref["depth"]) ** 2))),
            "mask_size": len(want), "mask_symmetric_difference": len(got ^ want),
            "mask_equal_outside_threshold_ulp": len(off_threshold) == 0, "tolerance": 1e-3}


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


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the parity check that uses its reference outputs")
    ap.add_argument("--no-chamfer", action="store_true", help="drop the ray chamfer loss term (runner.py:215-220; always on in the reference)")
    ap.add_argument("--no-flow", action="store_true", help="drop the scene-flow consistency loss (runner.py:222-253; the reference's default is on)")
    ap.add_argument("--chamfer", action="store_true", help=argparse.SUPPRESS)  # accepted for older command lines: now the default
    ap.add_argument("--flow", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sort-rays", action="store_true", help="serve the random pixels of a batch in 8x8-pixel-block order (locality experiment; measured slower)")
    ap.add_argument("--urf", action="store_true", help="add the line-of-sight loss (runner.py:255-276, opt.urf_loss)")
    ap.add_argument("--graph-staged", action="store_true", help="inference workloads: replay one captured hipGraph per chunk of the staged render (measured: no gain, a 4096-ray chunk is 3.4 ms of kernels)")
    ap.add_argument("--no-ema", action="store_true", help="no parameter EMA (the reference's default keeps one, --ema_decay 0.95, updated once per epoch)")
    ap.add_argument("--graph", action="store_true", help="training workloads: one hipGraph replay per step (Trainer.train_step_graphed) instead of eager launches")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)  # the default
    ap.add_argument("--settle-max", type=int, default=250, help="upper bound of the loss-scale settling loop before the warm-up (0: none; counter passes of the profiling script)")
    ap.add_argument("--profile-steps", type=int, default=2, help="steps of the per-kernel timing pass (0: no roofline block)")
    ap.add_argument("--variant-steps", type=int, default=10, help="timed steps of each secondary measurement (0: skip them)")
    ap.add_argument("--trained-steps", type=int, default=200, help="further training steps before the trained-state measurement")
    ap.add_argument("--flow-stream", dest="flow_stream", action="store_true", default=None, help="the scene-flow term on a stream of its own next to the render path (Trainer(flow_loss_stream=True))")
    ap.add_argument("--no-flow-stream", dest="flow_stream", action="store_false", help="... on the render path's stream")
    ap.add_argument("--force-dist", action="store_true", help="under torchrun with ONE rank: take the multi-rank code path anyway (RCCL init, barrier, two-phase gradient all-reduce) -- exercises the data-parallel path on a single-GPU box")
    ap.add_argument("--grad-transport", default="fp32", choices=("fp32", "bf16"), help="wire format of the hash-table / plane gradient ranges in the all-reduce (trainer.GradReducer)")
    ap.add_argument("--streams", type=int, default=None, help="l4d_streams_config mask (bit 0: encode in parts, 1: field adjoint forked, 2: static-grid pre-pass next to the LDS kernel); default: the library's")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce of the whole gradient arena behind the backward pass instead of two overlapped phases")
    return ap.parse_args()


PLUMBING = os.environ.get("L4D_BENCH_PLUMBING") == "1"  # tests/test_distributed_cpu.py: the launch / rendezvous / reduce / print
# path of this script on a machine without GPUs -- gloo instead of RCCL, a stand-in step, L4D_BENCH_FAKE_GPUS devices "present"


def relaunch_distributed(args):
    """--gpus N > 1 outside a torchrun environment: run N ranks of this script on this node."""
    n_dev = int(os.environ["L4D_BENCH_FAKE_GPUS"]) if PLUMBING and "L4D_BENCH_FAKE_GPUS" in os.environ else torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but this node exposes {n_dev} GPU(s); refusing to report a "
                         f"smaller run under that label")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env)
    if r.returncode != 0:
        raise SystemExit(f"bench.py: the {args.gpus}-rank launch failed with exit code {r.returncode}")

DETAIL_PATH = os.environ.get("L4D_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
LINE_LIMIT = 4096  # bytes of the ONE stdout line (the round-3 line had grown to 24 KB and the driver could not parse it)


def _r(x, n=4):
    return None if x is None else round(float(x), n)


def compact_line(detail, args):
    """The ONE stdout line: the contract's fields + scalar summaries (roofline of the dominant kernel, the hash-encoder figure,
    cpu_baseline, parity).  Everything else -- the per-kernel table, counter dictionaries, MFMA entries, prose -- is ``detail``
    and goes to bench_detail.json (L4D_BENCH_DETAIL) next to this script; its path is the line's ``detail`` field."""
    try:
        with open(DETAIL_PATH, "w") as f:
            json.dump(detail, f, indent=1)
        detail_ref = os.path.relpath(DETAIL_PATH, ROOT) if DETAIL_PATH.startswith(ROOT) else DETAIL_PATH
    except OSError as e:
        detail_ref = "not written: %r" % (e,)
    c = detail["config"]
    line = {k: detail[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                   "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 1), _r(line["ms_per_step"], 4)
    line["config"] = {"workload": c["workload"][:160], "rays_per_gpu_per_step": c["rays_per_gpu_per_step"], "samples_per_ray": c["samples_per_ray"],
                      "global_rays_per_step": c["global_rays_per_step"], "parallelism": c["parallelism"].split(",")[0],
                      "rccl_ranks_seen": c.get("rccl_ranks_seen"), "distinct_gpus_seen": c.get("distinct_gpus_seen"),
                      "skipped_steps_in_timed_region": c["skipped_steps_in_timed_region"], "skipped_steps_in_warmup": c["skipped_steps_in_warmup"],
                      "settling_steps": c["scaler_settling_steps_before_warmup"], "loss_scale": c["loss_scale_after_timed_region"],
                      "step_mode": c["step_mode"][:60]}
    ar = c.get("allreduce")
    if ar:  # multi-rank runs: bytes on the wire per phase and the part of the collective the step waited for (GradReducer.finish)
        line["config"]["allreduce"] = {"MB_early": _r(ar["bytes_early_phase"] / 1e6, 1), "MB_late": _r(ar["bytes_late_phase"] / 1e6, 1),
                                       "transport": ar["transport"], "exposed_wait_ms": _r(ar.get("exposed_wait_ms_per_step"), 3)}
    rf = detail.get("roofline")
    if rf:
        tr = rf.get("traffic") or {}
        req, miss = tr.get("l2_requests"), tr.get("l2_misses")
        alg = rf.get("algorithmic") or {}
        line["roofline"] = {"bound": rf["bound"], "kernel": rf["kernel"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
                            "achieved_from": str(rf.get("achieved_from"))[:60],
                            "traffic": tr.get("bytes_per_launch"), "frac_hbm_counter": rf.get("frac_hbm_counter"),
                            "algorithmic_GBps": alg.get("GBps"), "algorithmic_frac_of_l2_peak": alg.get("frac_of_l2_peak"),
                            "l2_miss_rate": _r(miss / req) if req and miss is not None else None,
                            "bytes_per_launch": rf["bytes_per_launch"], "avg_launch_ms": rf["avg_launch_ms"], "samples_per_launch": rf["samples_per_launch"],
                            "hash_gathers_G_per_s": rf.get("hash_gathers_G_per_s"), "gather_peak_G_per_s": GATHER_PEAK_G if rf.get("hash_gathers_G_per_s") else None,
                            "compulsory_hbm_frac": rf.get("compulsory_hbm_frac"),
                            "pmc_check_clean": (rf.get("pmc_check") or {}).get("clean")}
    else:
        line["roofline"] = None
    he = detail.get("hash_encoder")
    if he:
        line["hash_encoder"] = {"bound": "hbm", "peak": he["peak"], "unit": he["unit"], "target_frac": he["target_frac"], "ceiling_frac": he.get("ceiling_frac"),
                                **{k: {"frac": he[k]["frac"], "achieved": he[k]["achieved"], "ms": he[k]["ms"], "variant": he[k]["variant"]} for k in ("L8", "L16") if k in he}}
    mf = detail.get("mfma")
    if mf and mf.get("kernels"):
        top = max(mf["kernels"], key=lambda k: k["algorithmic_tflops"])
        util = [k["pmc_mfma_util_percent"] for k in mf["kernels"] if k.get("pmc_mfma_util_percent") is not None]
        line["mfma"] = {"peak_tflops": mf["peak_tflops"], "best_kernel": top["kernel"], "algorithmic_tflops": top["algorithmic_tflops"],
                        "frac": top["frac_of_dense_f16_peak"], "pmc_busy_percent_max": max(util) if util else None}
    v = detail.get("variants")
    if v:
        line["variants"] = {k: {"ms_per_step": _r(d["ms_per_step"], 3), "rays_per_s": _r(d["rays_per_s"], 0)} for k, d in v.items()}
        ts = v.get("trained_state")
        if ts:  # both operating points next to each other: the headline is the random-init state
            line["config"]["state"] = "random init (mask fraction ~1)"
            line["config"]["trained_state"] = {"ms_per_step": _r(ts["ms_per_step"], 3), "rays_per_s": _r(ts["rays_per_s"], 0),
                                               "mask_fraction": ts.get("mask_fraction"), "after_steps": ts.get("after_steps"),
                                               "zero_grad_row_fraction": ts.get("zero_grad_row_fraction"),
                                               "zero_grad_wave_fraction": ts.get("zero_grad_wave_fraction")}
    if "eval" in detail:
        line["eval"] = {"chamfer_f_score": detail["eval"]["chamfer_distance_m2, f_score@0.05"], "frames": detail["eval"]["frames"]}
    cb = detail.get("cpu_baseline")
    if cb:
        c1 = cb.get("c1_forward") or {}
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 2), "unit": cb.get("unit"), "cores": cb.get("cores"), "host_cores": cb.get("host_cores"),
                                "kind": cb.get("kind"), "sample": str(cb.get("sample"))[:240],
                                "c1_forward": {"value": _r(c1.get("value"), 1), "unit": c1.get("unit"), "frame_s": _r(c1.get("frame_s"), 1)} if c1 else None}
    pr = detail.get("parity")
    if pr:
        line["parity"] = {k: (pr[k] if not isinstance(pr[k], float) else float("%.3g" % pr[k])) for k in pr if k not in ("vs",)}
        if "error" in pr:
            line["parity"] = {"error": pr["error"][:200]}
    line["detail"] = detail_ref
    n = len(json.dumps(line))
    if n > LINE_LIMIT:  # never print a line the driver cannot parse: shed the optional blocks, biggest first
        for k in ("variants", "mfma", "eval", "parity", "hash_encoder"):
            if len(json.dumps(line)) <= LINE_LIMIT:
                break
            line[k] = "see " + detail_ref
        if len(json.dumps(line)) > LINE_LIMIT:  # config / roofline / cpu_baseline themselves are oversized: contract fields + pointer only
            keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data")
            line = {k: line[k] for k in keep if k in line}
            line["config"] = {"workload": str(detail.get("config", {}).get("workload", ""))[:200]}
            line["roofline"] = line["cpu_baseline"] = "see " + detail_ref
            line["detail"] = detail_ref
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_distributed(args)  # rank 0 of the child job prints the JSON line on our stdout
        return
    with _QuietStdout():
        line = _run(args)
    if line is not None:
        print(json.dumps(line), flush=True)


def timed(step, n, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    barrier()
    return time.perf_counter() - t0


def _run_plumbing(args, rank, world):
    """The multi-rank skeleton of _run with a stand-in step (a gradient-sized all-reduce on the CPU): same process-group
    set-up (gloo for RCCL), barrier, max-over-ranks timing and rank-0 JSON line.  What a first 8-GPU run can die on that is
    not a kernel."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    grad = torch.ones(1 << 16)
    n_rays = WORKLOADS[args.workload][1]

    def step():
        if world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
        time.sleep(0.001)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    dt = timed(step, args.steps, barrier)
    ranks_seen = dist.get_world_size() if world > 1 else 1
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
        dist.destroy_process_group()
    if rank != 0:
        return None
    return {"metric": "training rays/sec (64x1024 LiDAR panorama)", "value": n_rays * world * args.steps / dt, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none", "data": "synthetic",
            "config": {"workload": "PLUMBING TEST (L4D_BENCH_PLUMBING=1): stand-in step on the CPU over gloo, not a measurement", "parallelism": f"dp{world}",
                       "rccl_ranks_seen": ranks_seen, "distinct_gpus_seen": None}}


def _run(args):
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if PLUMBING:
        return _run_plumbing(args, rank, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # --force-dist: take the multi-rank code path (RCCL init, barrier, gradient all-reduce) even with one rank --
    # lets the data-parallel path be exercised on a single-GPU box (python -m torch.distributed.run --nproc-per-node 1)
    force_dist = args.force_dist and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from lidar4d_amd import LiDAR4D, _lib, ops
    from lidar4d_amd.data import KITTI360_SCALE, SyntheticKitti360
    from lidar4d_amd.trainer import Trainer

    model_kw, n_rays, desc = WORKLOADS[args.workload]
    if args.streams is not None:
        _lib.lib().l4d_streams_config(int(args.streams))
    torch.manual_seed(0)  # identical initial replicas on every rank
    model = LiDAR4D(near_lidar=1.0 * KITTI360_SCALE, far_lidar=81.0 * KITTI360_SCALE, num_frames=51, **model_kw).to(dev)
    inference = args.workload in INFERENCE
    data = SyntheticKitti360(dev, W=2048 if inference else 1024, num_rays=n_rays, seed=1000 + rank, sort_pixels=args.sort_rays,
                             frame_seed=1000)  # every rank: its own rays, the same frame sequence (equal work per step)
    use_chamfer, use_flow = not args.no_chamfer, not args.no_flow
    trainer = Trainer(model, data, chamfer=use_chamfer and not inference, flow=use_flow and not inference, urf=args.urf,
                      ema_decay=None if args.no_ema else 0.95, force_allreduce=force_dist, overlap_allreduce=not args.no_overlap,
                      grad_transport=args.grad_transport, flow_loss_stream=args.flow_stream is not False)
    if inference:
        from lidar4d_amd.data import KITTI360_FOV
        from lidar4d_amd.metrics import PointsMeter
        model.eval()
        model.graph_staged = args.graph_staged  # the 32 chunks of a frame replay one captured hipGraph (renderer.py)
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
    step_mode = "eager launches"

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if not inference and args.graph and not args.no_graph and trainer.graphs_supported():
        # One hipGraph per frame index, captured BEFORE the warm-up (a capture costs an eager step + the capture itself); the
        # training state those steps changed is put back afterwards, so the timed region starts where the eager loop would.
        opt, st = trainer.opt, model._store
        keep = {"flat": st.flat.detach().clone(), "m": opt.exp_avg.clone(), "v": opt.exp_avg_sq.clone(), "steps": opt.steps.clone(),
                "scaler": trainer.scaler.state.clone() if trainer.scaler is not None else None, "step_count": opt.step_count,
                "local_step": trainer.local_step, "ema": (trainer.ema.shadow.clone(), trainer.ema.num_updates) if trainer.ema is not None else None}
        try:
            t0 = time.perf_counter()
            for k in range(data.num_frames):
                trainer.train_step_graphed(k)
            torch.cuda.synchronize()
            capture_s = time.perf_counter() - t0
            from lidar4d_amd.params import bump_epoch
            with torch.no_grad():
                st.flat.copy_(keep["flat"]), opt.exp_avg.copy_(keep["m"]), opt.exp_avg_sq.copy_(keep["v"]), opt.steps.copy_(keep["steps"])
                if keep["scaler"] is not None:
                    trainer.scaler.state.copy_(keep["scaler"])
                opt.sched.copy_(torch.tensor([float(keep["step_count"]), 1.0]))
                if keep["ema"] is not None:
                    trainer.ema.shadow.copy_(keep["ema"][0])
                    trainer.ema.num_updates = keep["ema"][1]
            opt.step_count, trainer.local_step = keep["step_count"], keep["local_step"]
            bump_epoch()
            st.refresh16()
            step = trainer.train_step_graphed
            step_mode = f"one hipGraph replay per step (a graph per frame index, {data.num_frames} captured in {capture_s:.1f} s before the warm-up; state restored)"
        except Exception as e:  # the bench line must still be produced: fall back to eager launches
            step_mode = "eager launches (graph capture failed: %s)" % repr(e)[:200]
            sys.stderr.write("graph capture failed: %r\n" % (e,))
        del keep
    # GradScaler settling, outside every count: the reference's default init_scale (65536, on top of the internal 128) overflows
    # on some frames and the scale halves once per skipped step; a skipped step is a forward + backward without the Adam
    # update, i.e. not the step this benchmark is about.  The overflow threshold depends on the frame (its point clouds, its
    # time slices), so the loop runs until one epoch's worth of consecutive steps (51: every frame) was applied (<= 250
    # steps); then parameters, Adam moments, step counters and the EMA go back to their values at random init -- the timed
    # region starts from the state every earlier round measured -- and only the loss scale keeps what the loop found.
    settle_steps = 0
    if not inference and trainer.scaler is not None:
        from lidar4d_amd.params import bump_epoch
        st_, opt_ = model._store, trainer.opt
        snap = {"flat": st_.flat.detach().clone(), "m": opt_.exp_avg.clone(), "v": opt_.exp_avg_sq.clone(), "steps": opt_.steps.clone(),
                "count": opt_.step_count, "sched": None if opt_.sched is None else opt_.sched.clone(),
                "ema": None if trainer.ema is None else (trainer.ema.shadow.clone(), trainer.ema.num_updates),
                "global_step": getattr(trainer, "global_step", None)}
        applied_in_a_row, last = 0, int(opt_.steps.max())
        while applied_in_a_row < 51 and settle_steps < args.settle_max:
            step()
            settle_steps += 1
            now = int(opt_.steps.max())
            applied_in_a_row = applied_in_a_row + 1 if now > last else 0
            last = now
        with torch.no_grad():
            st_.flat.copy_(snap["flat"])
            opt_.exp_avg.copy_(snap["m"]), opt_.exp_avg_sq.copy_(snap["v"]), opt_.steps.copy_(snap["steps"])
            if snap["sched"] is not None and opt_.sched is not None:
                opt_.sched.copy_(snap["sched"])
        opt_.step_count = snap["count"]
        if snap["ema"] is not None:
            trainer.ema.shadow.copy_(snap["ema"][0])
            trainer.ema.num_updates = snap["ema"][1]
        if snap["global_step"] is not None:
            trainer.global_step = snap["global_step"]
        bump_epoch()  # fp16 compute copies, pair tables and the channel-last planes are rebuilt from the restored arena
        st_.refresh16()  # ... now: a graph replay does not contain the cast
        model.planes_encoder._cl_key = None
        del snap
    steps_before = int(trainer.opt.steps.max()) if not inference else 0
    for _ in range(args.warmup):
        step()
    steps_mid = int(trainer.opt.steps.max()) if not inference else 0
    reducer = getattr(trainer, "reducer", None)
    if reducer is not None:
        reducer.record_timing = True  # two events per step around GradReducer.finish: how much of the all-reduce was NOT hidden
    dt = timed(step, args.steps, barrier)
    allreduce = None
    if reducer is not None:
        reducer.record_timing = False
        allreduce = {"bytes_early_phase": reducer.bytes_early, "bytes_late_phase": reducer.bytes_late, "transport": reducer.transport,
                     "exposed_wait_ms_per_step": reducer.exposed_wait_ms()}
    # GradScaler: steps the device skipped (non-finite gradients while the scale backs off) are cheaper than real ones
    skipped = (args.steps - (int(trainer.opt.steps.max()) - steps_mid)) if not inference else None
    skipped_warmup = (args.warmup - (steps_mid - steps_before)) if not inference else None
    if world > 1 or force_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    scale_after = trainer.scaler.get_scale() if (not inference and trainer.scaler is not None) else None
    # what the process group really spans: ranks of the RCCL communicator, and how many DISTINCT devices they sit on
    # (all-gathered device UUIDs; two ranks on one GPU would show up here)
    ranks_seen, gpus_seen = (dist.get_world_size() if dist.is_initialized() else 1), 1
    if dist.is_initialized():
        uuid = str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")) or f"{socket.gethostname()}:{local_rank}"
        mine = torch.zeros(64, dtype=torch.uint8, device=dev)
        raw = uuid.encode()[:64]
        mine[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
        every = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(every, mine)
        gpus_seen = len({bytes(e.cpu().tolist()) for e in every})

    # ---- per-kernel timing pass (HIP events around every kernel launch, on the launch stream), outside the timed region ----
    roofline, roofline_kernels, mfma, per_step = None, None, None, {}
    if args.profile_steps > 0:
        # every rank runs the extra steps (they contain the gradient all-reduce); only rank 0 records events
        # eager launches (a graph replay does not pass the library's launch sites) with the side streams off: every kernel
        # is timed alone on the chip, which is what its roofline fraction refers to
        mask_was = ops.streams_mask()
        _lib.lib().l4d_streams_config(0)
        fls_was = getattr(trainer, "flow_loss_stream", False)
        trainer.flow_loss_stream = False  # (the scene-flow term back on the launch stream: its small kernels are timed alone, too)
        prof_step = step if inference else trainer.train_step
        if rank == 0:
            _lib.profile_start()
        for _ in range(args.profile_steps):
            prof_step()
        barrier()
        _lib.lib().l4d_streams_config(mask_was)
        trainer.flow_loss_stream = fls_was
    if rank == 0 and args.profile_steps > 0:
        kernels = {}
        for name, ms in _lib.profile_stop():
            kernels.setdefault(name, []).append(ms)
        per_step = {k: sum(v) / args.profile_steps for k, v in kernels.items()}
        P = min(n_rays, 4096) * 768 if inference else n_rays * 768  # staged inference launches per 4096-ray chunk
        with torch.no_grad():  # attribute rows = samples with weight > 1e-4 in the current state (one read-back, outside timing)
            b = data.batch_for(25) if not inference else None
            M = P if inference else int(model.render(b["rays_o_lidar"], b["rays_d_lidar"], b["time"], staged=False, perturb=True, num_steps=768)["mask_count"])
        models = kernel_models(model, P, M)
        if "hashgrid_t_fwd_levels_kernel<3, 8>" in kernels:  # the row kernel then only runs on the scene-flow loss's small point clouds:
            models.pop("hashgrid_t_fwd_kernel<3, 8, true>", None)  # its render-sized byte model does not apply to any launch of this step
        peaks = {"hbm": HBM_PEAK_GBS, "fabric": L2_PEAK_GBS, "lds": LDS_PEAK_GBS}  # ("fabric" rows: the algorithmic bytes against the L2 peak; the headline is the counter view below)
        # counter files of a committed rocprofv3 --pmc pass (tools/gpu_profile_round.sh): only attached when they were measured on
        # THIS build of the library (they carry its sha256) -- numbers of an older build would be constants, not measurements
        import hashlib
        lib_sha = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()

        def load_pmc(name):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path) or args.workload != "c3":
                return {}, None
            d = json.load(open(path))
            if d.get("library_sha256") != lib_sha:
                return {}, "dropped: measured on another build of the library (%s..., loaded %s...)" % (str(d.get("library_sha256"))[:12], lib_sha[:12])
            return d, "profiles/" + name

        def newest(pattern):  # the latest round's counter file that exists (tools/gpu_profile_round.sh <tag>)
            import glob
            c = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
            return os.path.basename(c[-1]) if c else pattern

        traffic_file, traffic_src = load_pmc(newest("hbm_traffic_r[0-9][0-9].json"))
        traffic = traffic_file.get("kernels", {})
        mfma_pmc, mfma_src = load_pmc(newest("r[0-9][0-9]_mfma_pmc.json"))
        def pmc_lookup(table, name):
            """profiles/ keys carry every template argument (rocprofv3's symbol), launch-site names only the explicit ones -- and the
            sorted scatter's launch macro leaves the compile-time bin size as the token DEF (binscatter.hip BS_LAUNCH): that is the
            instantiation whose last argument is not 0 (0 = run-time bin size)."""
            if name in table:
                return table[name]
            if name.endswith(", DEF>"):
                for k, v in table.items():
                    if k.startswith(name[:-len("DEF>")]) and not k.endswith(", 0>"):
                        return v
            for k, v in table.items():
                if name.endswith(">") and k.startswith(name[:-1] + ","):
                    return v
            return None

        rows, mf = [], []
        for name, times in kernels.items():
            row = {"kernel": name, "ms_per_step": round(per_step[name], 3), "launches_per_step": len(times) / args.profile_steps,
                   "avg_launch_ms": round(sum(times) / len(times), 4)}
            mod = models.get(name)
            if mod is not None:
                # some kernels also run on smaller inputs in the same step (the scene-flow loss evaluates the flow field on a
                # frame's point cloud: runner.py:227,252): the byte model is that of the render-sized launches, so only those
                # (duration within a factor two of the longest) are averaged
                big = [t for t in times if t >= 0.5 * max(times)]
                avg_ms = sum(big) / len(big)
                if mod.get("per_step"):  # a stage split over several launches of one kernel: modelled as one unit of work per step
                    big, avg_ms = [per_step[name]] * args.profile_steps, per_step[name]
                ach = mod["bytes"] / (avg_ms * 1e-3) / 1e9
                row.update(bound=mod["bound"], bytes_per_launch=mod["bytes"], modelled_launches_per_step=len(big) / args.profile_steps,
                           modelled_launch_ms=round(avg_ms, 4), achieved=round(ach, 1), peak=peaks[mod["bound"]], unit="GB/s",
                           frac=round(ach / peaks[mod["bound"]], 4), traffic=pmc_lookup(traffic, name), note=mod["note"])
                if row["traffic"]:  # memory-side bytes of the L2 (counters) over the measured duration, against the HBM peak
                    n_l = len(times) / args.profile_steps if mod.get("per_step") else 1.0  # (a stage of several launches: counters are per launch)
                    row["counter_bytes"] = row["traffic"]["bytes_per_launch"] * n_l
                    row["frac_hbm_counter"] = round(row["counter_bytes"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if "gathers" in mod:  # hash-entry gathers that are always issued (one lane = one table entry; coherent plane taps not counted)
                    gl = mod["gathers"] / (avg_ms * 1e-3) / 1e9
                    row["hash_gathers_G_per_s"] = round(gl, 1)
                    row["frac_of_measured_random_gather_rate"] = round(gl / GATHER_PEAK_G, 4)
                if "hbm" in mod:
                    row["compulsory_hbm_GBps"] = round(mod["hbm"] / (avg_ms * 1e-3) / 1e9, 1)
                    row["compulsory_hbm_frac"] = round(row["compulsory_hbm_GBps"] / HBM_PEAK_GBS, 4)
                if "flops" in mod:
                    tf = mod["flops"] / (avg_ms * 1e-3) / 1e12
                    pm = pmc_lookup(mfma_pmc.get("kernels", {}), name)
                    mf.append({"kernel": name, "algorithmic_tflops": round(tf, 1), "frac_of_dense_f16_peak": round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                               "pmc_mfma_util_percent": pm["mfma_util_percent"] if pm else None})
            rows.append(row)
        rows.sort(key=lambda r: -r["ms_per_step"])
        roofline_kernels = rows
        modelled = [r for r in rows if "bound" in r]
        # counter hygiene: every modelled kernel should carry counter traffic of THIS build, and the memory-side
        # bytes cannot be smaller than what the kernel must move once (hbm-bound models: compulsory bytes; gather kernels: their
        # compulsory streams) -- a violation means a wrong byte model or a wrong request-size conversion, and is reported, not hidden
        pmc_check = None
        if traffic:
            missing = [r["kernel"] for r in modelled if not r.get("traffic")]
            low = []
            for r in modelled:
                if r.get("traffic"):
                    must = r["bytes_per_launch"] if r["bound"] == "hbm" else models[r["kernel"]].get("hbm", 0)  # (gather kernels: their compulsory streams)
                    ratio = r.get("counter_bytes", r["traffic"]["bytes_per_launch"]) / max(must, 1)
                    r["counter_over_compulsory"] = round(ratio, 3)
                    if ratio < 0.95 and not models[r["kernel"]].get("upper_bound"):
                        low.append({"kernel": r["kernel"], "ratio": round(ratio, 3)})
            pmc_check = {"modelled_kernels": len(modelled), "without_traffic": missing, "counter_below_0.95x_compulsory": low,
                         "clean": not missing and not low}
        if modelled:
            d = modelled[0]  # the dominant modelled kernel
            fabric = d["bound"] == "fabric"
            # gather kernels: the headline is the fabric view -- counter bytes of THIS build / duration / 8 TB/s; without a
            # counter file of this build the compulsory HBM bytes stand in (a lower bound) and ``achieved_from`` says so
            if fabric and d.get("frac_hbm_counter") is not None:
                ach, src = d["counter_bytes"] / (d["modelled_launch_ms"] * 1e-3) / 1e9, "counter bytes (TCC_EA0 read requests by size + write requests) of this build / duration"
            elif fabric:
                ach, src = d.get("compulsory_hbm_GBps"), "compulsory HBM bytes / duration (no counter file of this build in profiles/: a lower bound of the fabric traffic)"
            else:
                ach, src = d["achieved"], {"hbm": "compulsory HBM bytes / duration", "lds": "table-entry bytes served from LDS / duration"}[d["bound"]]
            peak = HBM_PEAK_GBS if fabric else d["peak"]
            roofline = {"bound": d["bound"], "kernel": d["kernel"], "achieved": None if ach is None else round(ach, 1), "peak": peak, "unit": "GB/s",
                        "frac": None if ach is None else round(ach / peak, 4), "achieved_from": src,
                        "traffic": d["traffic"], "frac_hbm_counter": d.get("frac_hbm_counter"), "traffic_source": traffic_src,
                        "library_sha256": lib_sha, "bytes_per_launch": d["bytes_per_launch"],
                        "avg_launch_ms": d["modelled_launch_ms"], "launches_per_step": d["modelled_launches_per_step"],
                        "bytes_are": {"hbm": "compulsory HBM bytes", "fabric": "table-entry bytes gathered through L1/L2 (SURVEY 8d; tables are cache resident)",
                                      "lds": "table-entry bytes served from LDS"}[d["bound"]],
                        "algorithmic": {"bytes_per_launch": d["bytes_per_launch"], "GBps": d["achieved"], "frac_of_l2_peak": d["frac"], "l2_peak": L2_PEAK_GBS} if fabric else None,
                        "compulsory_hbm_GBps": d.get("compulsory_hbm_GBps"), "compulsory_hbm_frac": d.get("compulsory_hbm_frac"),
                        "hash_gathers_G_per_s": d.get("hash_gathers_G_per_s"),
                        "frac_of_measured_random_gather_rate": d.get("frac_of_measured_random_gather_rate"),
                        "gather_rate_note": "a gather costs one 128-byte line at the boundary it crosses: 264-292 G lane-loads/s chip-wide when the line is L2-resident "
                                            "(= the L2's 34 TB/s), 65 G/s when it comes over the fabric (= 8.3 TB/s), for 4 / 8 / 16 B per lane and every cache "
                                            "policy alike (tools/ubench/gather_policy.hip, profiles/r05_ubench_gather_policy.txt)",
                        "samples_per_launch": P, "attribute_rows": M, "pmc_check": pmc_check}
        mfma = {"kernels": mf, "peak_tflops": MFMA_F16_PEAK_TFLOPS,
                "pmc_source": (mfma_src + " (rocprofv3 --pmc pass of this command on this build: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs))") if mfma_pmc else mfma_src,
                "note": "algorithmic flops = the network's multiply-adds; the backward kernels execute about twice that on the matrix cores (operands are "
                        "produced in both orientations instead of being transposed through LDS), which is why their PMC utilisation is higher. "
                        "MLPs are 18-55 kFLOP per sample (SURVEY 8d): these kernels are bound by HBM streaming of their rows, not by the MFMA rate"}

    # ---- secondary measurements (single GPU): hash encoder alone, three-loss step, trained state ----
    hash_enc, variants = None, None
    if rank == 0 and world == 1 and not inference and args.variant_steps > 0:
        import numpy as np
        from lidar4d_amd.gridmeta import GridMeta
        b = data.batch_for(25)
        lin = torch.linspace(0.0, 1.0, 768, device=dev)
        noise = torch.rand(n_rays, 768, device=dev)
        _, xt = ops.sample_rays_xt(b["rays_o_lidar"].view(-1, 3).contiguous(), b["rays_d_lidar"].view(-1, 3).contiguous(), lin, noise,
                                   b["time"].view(-1), float(np.float32(model.near_lidar)), float(np.float32(model.far_lidar)), model.bound)
        hash_enc = {"what": "static 3-D hash grid forward alone (l4d_hashgrid_fwd / l4d_hashgrid_fwd_ws), ray-ordered samples of one batch, F = 4, 2^19-entry tables, fp16; "
                            "rows_kernel: one thread per point, all levels; level_major: one level at a time chip-wide (every L2 holds that level's table), "
                            "x-neighbour pairs in one 16-byte load where aligned, level-major scratch, row assembly (both kernels timed)",
                    "samples": xt.shape[0], "peak": HBM_PEAK_GBS, "unit": "GB/s", "target_frac": 0.40,
                    # every gather occupies the address path for ~2.1 clocks per distinct address of the instruction: 292 G lane-loads/s when
                    # every line is L2-resident (tools/ubench/gather.hip); with pair loads a level's 8 entries cost 6 address slots
                    "ceiling_frac": round(GATHER_PEAK_G * 8.0 * (8.0 / 6.0) / HBM_PEAK_GBS, 4),
                    "ceiling_note": "8-byte entry gathers are bound by the address path (292 G distinct-address slots/s all-hit, 65 G/s over the fabric; "
                                    "profiles/r02_ubench_gather.txt, r05_ubench_gather_policy.txt), not by bytes; a level's 8 corners take 6 slots with "
                                    "x-neighbour pair loads: ceiling 292 G/s x 8 B x 8/6 = 3.1 TB/s = 0.39 of the HBM peak if every lane's line were L2-resident "
                                    "and the coarse levels cost what the fine ones do"}
        for Lh in (8, 16):
            meta = GridMeta(3, Lh, 4, 19, 512, np.exp2(np.log2(32768 / 512) / (Lh - 1)))
            table = ((torch.rand(meta.n_params, device=dev) - 0.5)).half()
            entry = {"algorithmic_bytes_per_sample": Lh * 64}
            for tag, lm in (("rows_kernel", False), ("level_major", True)):
                out = ops.hashgrid_fwd(meta, xt, (0, 1, 2), table, level_major=lm)
                torch.cuda.synchronize()
                _lib.profile_start()
                for _ in range(5):
                    ops.hashgrid_fwd(meta, xt, (0, 1, 2), table, out=out, level_major=lm)
                ms = sum(v for _, v in _lib.profile_stop()) / 5
                gbs = Lh * 64 * xt.shape[0] / (ms * 1e-3) / 1e9
                entry[tag] = {"ms": round(ms, 4), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
                del out
            best = max(("rows_kernel", "level_major"), key=lambda t: entry[t]["frac"])
            entry.update(ms=entry[best]["ms"], achieved=entry[best]["achieved"], frac=entry[best]["frac"], variant=best)
            hash_enc[f"L{Lh}"] = entry
            del table
        variants = {}
        if use_chamfer or use_flow:
            trainer.chamfer = trainer.flow = False
            for _ in range(2):
                trainer.train_step()
            dtv = timed(trainer.train_step, args.variant_steps, barrier)  # eager launches (the captured graphs hold the full step)
            variants["three_loss_step"] = {"step": "L1 depth + MSE raydrop + MSE intensity only (no ray chamfer, no scene-flow loss)", "steps": args.variant_steps,
                                           "ms_per_step": dtv / args.variant_steps * 1e3, "rays_per_s": n_rays * args.variant_steps / dtv}
            trainer.chamfer, trainer.flow = use_chamfer, use_flow
        if args.trained_steps > 0:
            for _ in range(args.trained_steps):
                step()
            dtv = timed(step, args.variant_steps, barrier)
            with torch.no_grad():
                bb = data.batch_for(25)
                frac = float(model.render(bb["rays_o_lidar"], bb["rays_d_lidar"], bb["time"], staged=False, perturb=True, num_steps=768)["mask_count"]) / (n_rays * 768)
            # exact zeros of the adjoint: rows of dh = d loss / d (sigma-network output) that are all-zero fp16 after the
            # compositing and attribute backward produce no gradient anywhere downstream; measured on one extra step, per row, per
            # 64-row wavefront segment and per 512-row tile (what a wave- / workgroup-uniform skip would save)
            zero_rows = {}

            def _probe(dh):
                z = (dh == 0).all(dim=1)
                zero_rows.update(rows=float(z.float().mean()), waves=float(z.view(-1, 64).all(dim=1).float().mean()),
                                 tiles=float(z.view(-1, 512).all(dim=1).float().mean()))
            model._bwd_probe = _probe
            trainer.train_step()
            torch.cuda.synchronize()
            model._bwd_probe = None
            variants["trained_state"] = {"zero_grad_row_fraction": round(zero_rows.get("rows", -1.0), 4),
                                         "zero_grad_wave_fraction": round(zero_rows.get("waves", -1.0), 4),
                                         "zero_grad_tile_fraction": round(zero_rows.get("tiles", -1.0), 4),
                                         "after_steps": args.warmup + args.steps + args.profile_steps + args.trained_steps + 2 * args.variant_steps,
                                         "what": f"the headline step after {args.warmup + args.steps + args.profile_steps + args.trained_steps + 2 * args.variant_steps} training "
                                                 "steps on the synthetic scene: fewer samples pass the weights > 1e-4 mask (attribute networks run on those only), flow gradients are no longer tiny",
                                         "steps": args.variant_steps, "ms_per_step": dtv / args.variant_steps * 1e3, "rays_per_s": n_rays * args.variant_steps / dtv,
                                         "mask_fraction": round(frac, 4), "loss_scale": trainer.scaler.get_scale() if trainer.scaler is not None else None}

    line = None
    if rank == 0:
        total_rays = n_rays * world * args.steps
        losses = "L1 depth + MSE raydrop + MSE intensity" + (" + ray chamfer" if use_chamfer else "") + \
                 (" + scene-flow consistency" if use_flow else "") + (" + line-of-sight" if args.urf else "")
        detail = {
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
            "dtype": "f16",
            "dtype_note": "f16 tables / MFMA operands, f32 accumulate, f32 planes and compositing",
            "data": "synthetic",
            "config": {"workload": desc, "rays_per_gpu_per_step": n_rays, "samples_per_ray": 768,
                       "global_rays_per_step": n_rays * world,
                       "parallelism": f"frame-sharded x{world}, no collective" if inference else f"ray-sharded dp{world}, one RCCL gradient all-reduce per step in two phases, overlapped with the tail of the backward pass",
                       "step": "no_grad render(staged=True, max_ray_batch=4096) + U-Net + pano_to_lidar + chamfer/F-score" if inference else
                               f"the reference's training step (runner.py:166-253,474-551): forward + backward + GradScaler check/skip/update + Adam; losses {losses}; "
                               + ("no parameter EMA" if args.no_ema else "parameter EMA once per epoch of 51 steps (runner.py:534-535)"),
                       "state": "random init (tiny-cuda-nn default U(-1e-4, 1e-4) tables): every sample passes the weights > 1e-4 mask, the attribute networks run on all of them",
                       "loss_scale_after_timed_region": scale_after, "skipped_steps_in_timed_region": skipped,
                       "skipped_steps_in_warmup": skipped_warmup, "scaler_settling_steps_before_warmup": settle_steps, "step_mode": step_mode,
                       "side_streams_mask": ops.streams_mask(),
                       "scene_flow_term_on_its_own_stream": bool(getattr(trainer, "flow_loss_stream", False)) and not inference and use_flow and step_mode.startswith("eager"),
                       "rccl_ranks_seen": ranks_seen, "distinct_gpus_seen": gpus_seen, "allreduce": allreduce},
            "roofline": roofline,
            "roofline_kernels": roofline_kernels,
            "mfma": mfma,
            "hash_encoder": hash_enc,
            "variants": variants,
        }
        if inference:
            detail["eval"] = {"chamfer_distance_m2, f_score@0.05": [float(v) for v in meter.measure()], "frames": meter.N,
                              "note": "random-init field vs synthetic ground truth: exercises the eval path, not a quality claim"}
        if world == 1 and not args.no_cpu_baseline and not inference:
            with tempfile.TemporaryDirectory() as td:
                ref_path = os.path.join(td, "parity_ref.npz")
                detail["cpu_baseline"] = cpu_baseline(51, KITTI360_SCALE, ref_path)
                try:
                    detail["parity"] = parity_against(ref_path, dev, KITTI360_SCALE) if os.path.exists(ref_path) else None
                except Exception as e:  # the bench line must still be produced
                    detail["parity"] = {"error": repr(e)[:300]}
        line = compact_line(detail, args)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
