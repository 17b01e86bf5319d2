"""Turn the --json dumps of tools/rocpd_pmc.py (separate rocprofv3 --pmc passes of `bench.py`) into the two small files
bench.py attaches to its JSON line:

    python tools/pmc_to_profiles.py <fetch.json> <write.json> <mfma.json> <out_dir> <tag>

  <out_dir>/hbm_traffic_<tag>.json   per kernel: fabric-side bytes per launch = FETCH_SIZE x 1024 x c + WRITE_SIZE x 1024, with
                                 c = 2 for streaming kernels and c = 1 for gather kernels -- calibration (tools/gather_calib.py,
                                 profiles/<tag>_fetch_size_calibration.txt): FETCH_SIZE reports HALF of a wide coalesced stream
                                 (the guide's rule, reproduced) but 64 B per random 8-byte gather, i.e. it must NOT be doubled
                                 for the gather kernels.  The counters sit at the L2 <-> fabric boundary: Infinity-Cache hits are
                                 included, so this is an upper bound on HBM traffic.
  <out_dir>/<tag>_mfma_pmc.json      per MLP kernel: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) -- on this
                                 8-XCD part rocprofv3 sums GRBM_GUI_ACTIVE over the XCDs (a 2.1 ms kernel reports 3.8e7 cycles = 8 x
                                 4.7e6), so the stock derived metric, which divides by the raw sum, under-reports by 8x -- and the
                                 executed MFMA flops SQ_INSTS_VALU_MFMA_MOPS_F16 x 512.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_pmc import simple_demangle

GATHER = ("density_encode_fwd_kernel", "hashgrid_t_fwd_kernel", "hashgrid_fwd_kernel", "field_bwd_prep_kernel", "planes_dyn_lds_kernel",
          "attr_gather_kernel", "attr_gather_bwd")


def main(fetch, write, mfma, out_dir, tag):
    F, W, M = ({simple_demangle(k): v for k, v in json.load(open(p)).items()} for p in (fetch, write, mfma))
    traffic = {}
    for k in sorted(set(F) | set(W)):
        f = F.get(k, {}).get("FETCH_SIZE")
        w = W.get(k, {}).get("WRITE_SIZE")
        if not f and not w:
            continue
        c = 1 if k.startswith(GATHER) else 2
        fb = f["sum"] / f["launches"] * 1024 * c if f else 0.0
        wb = w["sum"] / w["launches"] * 1024 if w else 0.0
        traffic[k] = {"bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb, "fetch_size_factor": c,
                      "launches": (f or w)["launches"]}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 2 --warmup 1, workload c3",
               "note": "fabric-side (L2 miss) traffic incl. Infinity-Cache hits; FETCH_SIZE x 2 only for streaming kernels (calibrated)",
               "kernels": traffic}, open(f"{out_dir}/hbm_traffic_{tag}.json", "w"), indent=1)
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
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 of bench.py --steps 2 --warmup 1, workload c3",
               "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); 100 % = every SIMD issuing MFMAs back to back = 2.5 PFLOP/s f16",
               "kernels": util}, open(f"{out_dir}/{tag}_mfma_pmc.json", "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:6])
