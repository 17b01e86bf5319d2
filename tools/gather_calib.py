"""Calibration of rocprofv3's FETCH_SIZE on RANDOM gathers of 8 / 16 / 32 bytes (the guide calibrates it on wide coalesced
streaming reads only, where it reports exactly half of the bytes).  The table (1 GiB) is four times the Infinity Cache, so
nearly every gather is an HBM miss of one cache line; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the
gather kernels' FETCH_SIZE (KB) per launch with  n_gathers x line size (+ the streamed index bytes).
"""
import sys
import torch

dev = "cuda"
n_idx = 1 << 24
g = torch.Generator(device=dev).manual_seed(0)
for width in (1, 2, 4):  # int64 columns per row: 8, 16, 32 bytes per gathered element
    rows = (1 << 30) // (8 * width)
    table = torch.zeros(rows, width, dtype=torch.int64, device=dev)
    idx = torch.randint(0, rows, (n_idx,), device=dev, generator=g)
    out = torch.index_select(table, 0, idx)   # warm-up of the kernel selection
    torch.cuda.synchronize()
    for _ in range(3):
        out = torch.index_select(table, 0, idx)
    torch.cuda.synchronize()
    print(f"width {8 * width} B: {n_idx} random gathers from {rows} rows (1 GiB table); index stream {n_idx * 8 / 1e6:.0f} MB, "
          f"output {n_idx * 8 * width / 1e6:.0f} MB; one 64-B line per gather = {n_idx * 64 / 1e6:.0f} MB, 128-B = {n_idx * 128 / 1e6:.0f} MB")
    del table, idx, out
# a streaming copy of known size for the coalesced reference point
src = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
print(f"streaming copy: {src.numel() * 4 / 1e6:.0f} MB read, same written")
