"""world_size-2 gloo test of the data-parallel path (bench.py --gpus N / lidar4d_amd.trainer.Trainer): each rank
renders its own ray shard, the flat gradient arena is SUM-all-reduced once, and the result equals the gradient of
one process rendering the concatenated batch.  Runs on CPU: the render itself is the oracle (the HIP path needs a
GPU); what is under test is the sharding + flat-buffer all-reduce + optimizer-group layout logic."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.make_golden import SMALL_MODEL


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lidar4d_amd import LiDAR4D
    from lidar4d_amd.trainer import lidar_loss
    from oracle import fields_ref, tcnn_ref
    from oracle.detparams import det_uniform, fill_model
    from oracle.make_golden import test_rays
    tcnn_ref.set_precision("fp32")
    cfg = dict(SMALL_MODEL, density_scale=40.0)
    ref = fill_model(fields_ref.LiDAR4D(**cfg), seed=7)      # compute stand-in for the (GPU-only) HIP model
    shell = fill_model(LiDAR4D(**cfg), seed=7)               # the product's parameter store / grad arena
    store = shell._store
    n_total, steps = 16, 32
    ro, rd = test_rays(n_total, 5)
    noise = det_uniform((n_total, steps), "dn", 0, 1)
    images = torch.stack([(det_uniform((1, n_total), "ia", 0, 1) > 0.2).float(), det_uniform((1, n_total), "ib", 0, 1),
                          det_uniform((1, n_total), "ic", 0.01, 0.8)], -1)
    t = torch.tensor([[0.5]])

    def grads_for(lo, hi):
        ref.zero_grad()
        o = ref.render(ro[:, lo:hi], rd[:, lo:hi], t, num_steps=steps, perturb=True, noise=noise[lo:hi])
        lidar_loss(o, images[:, lo:hi]).backward()
        g = store.prepare_grads()
        g.zero_()
        store.gates[rank + 1] = 1.0  # "this rank's step used time slice rank + 1": the gates ride in the arena's tail
        pr = dict(ref.named_parameters())
        for name, p, off, n, _ in store.entries:
            if n and pr[name].grad is not None:
                g[off:off + n] = pr[name].grad.reshape(-1)
        return g

    shard = n_total // world

    def same(a, b):
        """two ranks: a + b in either order is the same float, so every way of cutting the arena into all-reduces agrees bit for
        bit; from three ranks on the order in which an all-reduce algorithm adds the ranks' values depends on where an element
        sits in the buffer it was handed: equal to fp32 rounding of a 4-term sum"""
        if world == 2:
            return bool(torch.equal(a, b))
        return bool((a - b).abs().max() <= 4e-7 * b.abs().max())

    g = grads_for(rank * shard, (rank + 1) * shard).clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)  # what Trainer.train_step does on flat_grad
    # the trainer's two-phase (overlapped) reduction must give exactly what one all-reduce of the arena gives
    from lidar4d_amd.trainer import GradReducer
    red = GradReducer(shell)
    mine = grads_for(rank * shard, (rank + 1) * shard)   # = store.flat_grad, this rank's shard gradient again
    assert mine.data_ptr() == store.flat_grad.data_ptr()
    red.early()      # what the fused backward triggers once the non-flow gradients are final
    red.finish()     # flow range + wait
    two_phase_equal = same(store.flat_grad, g)
    mine = grads_for(rank * shard, (rank + 1) * shard)
    red.finish()     # no early(): single all-reduce fallback
    fallback_equal = same(store.flat_grad, g)
    # bf16 transport of the encoder range (all-to-all, fp32 sum on arrival, all-gather): the fp32 result rounded to bf16 once
    red16 = GradReducer(shell, transport="bf16")
    # expected: every rank's contribution rounded to bf16, summed in fp32 in rank order, the sum rounded to bf16 once
    mine = grads_for(rank * shard, (rank + 1) * shard)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    want16 = sum(p_[:red16.flow_lo].to(torch.bfloat16).float() for p_ in parts).to(torch.bfloat16).float()
    for two_phase in (True, False):
        mine = grads_for(rank * shard, (rank + 1) * shard)
        if two_phase:
            red16.early()
        red16.finish()
        lo = red16.flow_lo
        bf16_ok = bool(torch.equal(store.flat_grad[:lo], want16)) and same(store.flat_grad[lo:], g[lo:])  # (the bf16 range: rank order, exact)
        fallback_equal = fallback_equal and bf16_ok
    if rank == 0:
        full = grads_for(0, shard * world)
        n = store.numel  # parameter gradients only (the tail behind them holds the gates)
        err = float((g[:n] - full[:n]).abs().max() / full[:n].abs().max())
        uneven = red16.flow_lo % world != 0
        out.put(("err", err, int(store.numel), [list(r) for r in store.group_ranges], two_phase_equal, fallback_equal,
                 [red.flow_lo, red.flow_hi], g[store.numel:store.numel + 4].tolist(), int(full.numel()), uneven))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ray_sharded_allreduce_equals_single_batch(world):
    """world 2; world 4: four shards of four rays; world 3: the bf16 transport's all-to-all with chunks that do
    not divide the encoder range evenly (ceil(n / 3) elements per rank, zero-padded tail -- the range is a multiple of 4 and 8,
    so three ranks is where that path runs)."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    tag, err, numel, ranges, two_phase_equal, fallback_equal, flow_range, gates, grad_numel, uneven = out.get(timeout=10)
    assert uneven == (world == 3)
    assert gates == [0.0] + [1.0] * min(world, 3) + [0.0] * (3 - min(world, 3)) and grad_numel == numel + 32  # the SUM all-reduce merges the ranks' slice gates
    assert tag == "err" and err < 1e-5, err
    assert two_phase_equal and fallback_equal
    assert ranges[1][0] == flow_range[0] < flow_range[1] < numel  # the flow field opens lr group 1
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == numel  # two contiguous lr groups


def test_bench_multi_rank_launch_plumbing(tmp_path):
    """bench.py --gpus 2 outside a torchrun environment: relaunch_distributed spawns two ranks through torch.distributed.run
    on 127.0.0.1, they rendezvous, all-reduce, agree on the max-over-ranks time, and exactly ONE JSON line with n_gpus = 2
    reaches stdout (rank 0's) -- the path the driver's first multi-GPU run takes, here with gloo, a stand-in step and two
    pretended devices (L4D_BENCH_PLUMBING=1: no GPU in this container)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, L4D_BENCH_PLUMBING="1", L4D_BENCH_FAKE_GPUS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    assert len(lines[0]) < 4096  # what the driver can parse
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rccl_ranks_seen"] == 2  # the size of the process group the ranks really formed, next to the requested n_gpus
    # the driver's largest launch: 8 ranks (rendezvous, port, MAX-over-ranks reduction and the single line must survive it)
    env["L4D_BENCH_FAKE_GPUS"] = "8"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks_seen"] == 8 and d["value"] > 0
    # more ranks than devices: refused loudly, nothing that looks like a result on stdout
    env["L4D_BENCH_FAKE_GPUS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=str(tmp_path))
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and not r.stdout.strip().startswith("{")


def test_bench_line_is_compact(tmp_path, monkeypatch):
    """the ONE stdout line of bench.py must stay parseable by the driver.  ``compact_line`` is fed the
    largest detail record any round produced (round 3's 24 KB line) and must return < 4 KB of JSON that round-trips and
    still carries the contract's fields, ``roofline`` (scalars only) and ``cpu_baseline``; the detail goes to the side file."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    detail = json.load(open(os.path.join(root, "profiles", "r03_bench_c3_final.json")))
    assert len(json.dumps(detail)) > 20000
    line = bench.compact_line(detail, None)
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT == 4096
    back = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity", "hash_encoder"):
        assert k in back, k
    assert back["config"]["workload"].startswith("C3") and back["config"]["skipped_steps_in_timed_region"] == 0
    rf = back["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in rf.values())  # scalars only
    assert rf["frac"] == detail["roofline"]["frac"] and rf["traffic"] > 0 and 0 < rf["l2_miss_rate"] < 1
    assert back["cpu_baseline"]["kind"] == "port" and back["cpu_baseline"]["cores"] == 32
    assert json.load(open(tmp_path / "bench_detail.json"))["roofline_kernels"] == detail["roofline_kernels"]
