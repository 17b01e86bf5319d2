"""Counter summaries.  Turns the --json dumps of tools/rocpd_pmc.py (separate rocprofv3 --pmc passes of `bench.py`)
into the files bench.py attaches to its detail record -- tied to the library build they were measured on:

    python tools/pmc_summary.py <rd.json> <wr.json> <l2.json> <mfma.json> <out_dir> <tag> <path to liblidar4d_hip.so>

Round 4: gfx950 counts the memory-side read requests of the L2 BY SIZE (TCC_EA0_RDREQ_32B / _64B / _128B, counter_defs.yaml),
so read bytes = 32 x n32 + 64 x n64 + 128 x n128 exactly; the per-kernel "access class" guess of round 3 (kept as a fallback
for dumps without the size counters) is gone from the numbers -- it had classed the MLP kernels' row streams as 64-byte
requests and reported fewer bytes than the rows they read.

<out_dir>/hbm_traffic_<tag>.json   per kernel and launch, from the L2's memory-side request counters (the guide's FETCH_SIZE /
    WRITE_SIZE are derived from the same counters with a formula that assumes 64-byte requests):
      read_requests  = TCC_EA0_RDREQ_sum            (TCC_EA0_RDREQ_32B_sum of them 32-byte: none on this workload)
      read_bytes     = read_requests x 64 or x 128 by the kernel's access class -- calibrated with tools/ubench/calib.hip
                       (profiles/<tag>_counter_calibration.txt): a wave-level load that covers >= 128 contiguous bytes (4-, 12-,
                       16-byte lanes alike) goes out as ONE 128-byte request per 128 bytes, a load whose lanes touch separate
                       64-byte sectors (16 B out of a 256-byte row, the four 16-byte pieces of one 64-byte quarter of a row,
                       random 8-byte gathers) as one 64-byte request per sector.  Both bounds are kept (read_bytes_64, _128).
      dram_fraction  = TCC_EA0_RDREQ_DRAM_sum / TCC_EA0_RDREQ_sum: the share of those requests that went on to HBM (the rest
                       was served by the Infinity Cache)
      write_bytes    = TCC_EA0_WRREQ_64B_sum x 64 + (TCC_EA0_WRREQ_sum - TCC_EA0_WRREQ_64B_sum) x 32
      l2_hit_rate    = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), l2_requests = TCC_REQ_sum
<out_dir>/<tag>_mfma_pmc.json      per MLP kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
Both carry "library_sha256": bench.py drops the counter fields when the library it loaded is a different build.
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_pmc import simple_demangle

# access class of a kernel's dominant READ stream (see the docstring): "stream" = 128-byte requests, "sector" = 64-byte requests.
# Kernels that mix both (a streamed index or coordinate array next to gathers) are classed by where most requests come from.
STREAM = ("planes_static_lds_kernel", "dynhash_lds_kernel", "dynhash_fwd_lds_kernel", "adam_ranges_kernel", "composite_", "sample_rays",
          "sigma_bwd_kernel", "nonfinite_check_kernel", "warp_coords_kernel", "cast_kernel", "relayout_kernel", "chamfer_", "dyn_pairs_kernel")


def per_launch(d, name):
    """average per launch over the kernel's LARGE launches when the dump separates them (rocpd_pmc.py big_*), else over all"""
    c = d.get(name)
    if not c or not c.get("launches"):
        return None
    if c.get("big_launches"):
        return c["big_sum"] / c["big_launches"]
    return c["sum"] / c["launches"]


def main(rd, wr, l2, mfma, out_dir, tag, lib):
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    R, W, L, M = ({simple_demangle(k): v for k, v in json.load(open(p)).items()} for p in (rd, wr, l2, mfma))
    traffic = {}
    for k in sorted(set(R) | set(W)):
        req = per_launch(R.get(k, {}), "TCC_EA0_RDREQ_sum")
        if req is None and k not in W:
            continue
        req = req or 0.0
        r32 = per_launch(R.get(k, {}), "TCC_EA0_RDREQ_32B_sum") or 0.0
        dram = per_launch(R.get(k, {}), "TCC_EA0_RDREQ_DRAM_sum")
        if dram is None:  # round 4: counted in the write pass (four TCC slots per pass)
            dram = per_launch(W.get(k, {}), "TCC_EA0_RDREQ_DRAM_sum")
        wreq = per_launch(W.get(k, {}), "TCC_EA0_WRREQ_sum") or 0.0
        w64 = per_launch(W.get(k, {}), "TCC_EA0_WRREQ_64B_sum") or 0.0
        n64 = per_launch(R.get(k, {}), "TCC_EA0_RDREQ_64B_sum")
        n128 = per_launch(R.get(k, {}), "TCC_EA0_RDREQ_128B_sum")
        rb64, rb128 = (req - r32) * 64 + r32 * 32, (req - r32) * 128 + r32 * 32
        if n64 is not None and n128 is not None:  # exact: requests counted by size
            cls = "by_size"
            rb = r32 * 32 + n64 * 64 + n128 * 128 + max(req - r32 - n64 - n128, 0.0) * 64
        else:
            cls = "stream" if k.startswith(STREAM) else "sector"
            rb = rb128 if cls == "stream" else rb64
        wb = w64 * 64 + (wreq - w64) * 32
        e = {"read_requests": req, "read_requests_32B": r32, "read_requests_64B": n64, "read_requests_128B": n128,
             "read_bytes_64": rb64, "read_bytes_128": rb128, "access_class": cls, "fetch_bytes": rb,
             "write_bytes": wb, "bytes_per_launch": rb + wb, "dram_fraction": None if dram is None or req == 0 else round(dram / req, 4),
             "launches": (R.get(k) or W.get(k))[next(iter(R.get(k) or W.get(k)))]["launches"]}
        hit, miss, lreq = (per_launch(L.get(k, {}), c) for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"))
        if hit is not None and miss is not None and hit + miss > 0:
            e.update(l2_hit_rate=round(hit / (hit + miss), 4), l2_requests=lreq, l2_misses=miss)
        traffic[k] = e
    json.dump({"library_sha256": sha,
               "source": "rocprofv3 --pmc (TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum | TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum | "
                         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum), separate passes of bench.py --steps 2 --warmup 1 (side streams off), workload c3",
               "note": "memory-side requests of the L2 (Infinity-Cache hits included; dram_fraction = share that reached HBM); request size by access class, "
                       "calibrated with tools/ubench/calib.hip", "kernels": traffic}, open(f"{out_dir}/hbm_traffic_{tag}.json", "w"), indent=1)
    util = {}
    for k, c in M.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] == 0:
            continue
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"]
        gui = c["GRBM_GUI_ACTIVE"]["sum"] / c["GRBM_GUI_ACTIVE"]["launches"]
        util[k] = {"mfma_busy_cycles_per_launch": busy, "gui_active_cycles_per_launch_sum_over_8_xcds": gui,
                   "mfma_util_percent": round(100.0 * busy / (gui / 8.0 * 1024.0), 2)}
        if "SQ_INSTS_VALU_MFMA_MOPS_F16" in c:
            util[k]["executed_mfma_flops_per_launch"] = c["SQ_INSTS_VALU_MFMA_MOPS_F16"]["sum"] / c["SQ_INSTS_VALU_MFMA_MOPS_F16"]["launches"] * 512
    json.dump({"library_sha256": sha,
               "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA of bench.py --steps 2 --warmup 1, workload c3",
               "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); 100 % = every SIMD issuing MFMAs back to back = 2.5 PFLOP/s f16",
               "kernels": util}, open(f"{out_dir}/{tag}_mfma_pmc.json", "w"), indent=1)
    print(f"wrote {out_dir}/hbm_traffic_{tag}.json ({len(traffic)} kernels) and {out_dir}/{tag}_mfma_pmc.json ({len(util)} kernels), library {sha[:16]}")


if __name__ == "__main__":
    main(*sys.argv[1:8])
