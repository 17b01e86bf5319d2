#!/usr/bin/env python3
"""Bottom-up floor of the training step: per kernel, the time its BINDING roof would allow, next to the time it takes.

    python tools/floor_table.py <bench_detail.json> <pmc_inst.json> [<hbm_traffic.json>] [--md]

Roofs (all measured on this chip):
  issue   every wave-level instruction needs an issue slot of its SIMD.  SQ_INSTS_VALU / SALU / LDS / VMEM / SMEM (rocprofv3 --pmc,
          tools/gpu_pmc_quick.sh "inst") are the kernel's dynamic instruction counts; every VALU instruction of a wave64 occupies
          its SIMD for one quad-cycle = ISSUE_CYCLES cycles (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 quad-cycles for
          every kernel of this step, profiles/r04_pmc_sq.txt + r04_pmc_inst.txt; packed fp32 buys nothing)
          -> t_issue = INSTS_VALU x ISSUE_CYCLES / (1024 SIMDs x CLOCK).
          MFMA kernels: + SQ_INSTS_MFMA x 16 cycles (v_mfma_f32_16x16x32_f16: 4 passes of 4 cycles) on the matrix pipe, which
          runs beside the VALU -> max of the two.
  hbm     compulsory bytes of the byte model (bench.py kernel_models) / 6.3 TB/s achievable.
  miss    gather kernels: lines that miss an XCD's L2 arrive at 64 G lines/s (TCC_MISS of the traffic file).
  gather  gather kernels: 292 G distinct-address slots/s when every line hits (table-entry gathers only: plane taps not counted).
  gather+miss  the sum of the two (round 5: they add).
floor = max of the applicable ones; gap = time - floor.  The sum of the floors is what THIS design (this instruction stream, this
traffic) could reach if every kernel sat on its roof -- not a lower bound for the problem.
"""
import json
import sys

CLOCK = 2.3e9          # sustained shader clock under these kernels (GRBM_GUI_ACTIVE / wall, profiles/*_pmc_mfma)
SIMDS = 1024
ISSUE_CYCLES = 4.0     # one wave64 VALU instruction per quad-cycle per SIMD (measured, see above)
MFMA_CYCLES = 16.0     # v_mfma_f32_16x16x32_f16 on one SIMD
HBM = 6.3e12
MISS_LINES = 64e9
GATHER = 292e9


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    detail = json.load(open(args[0]))
    inst = json.load(open(args[1]))
    traffic = json.load(open(args[2]))["kernels"] if len(args) > 2 else {}
    md = "--md" in sys.argv

    def pmc(name, counter):
        for k, v in inst.items():
            if k == name or (name.endswith(">") and k.startswith(name[:-1] + ",")) or (name.endswith(", DEF>") and k.startswith(name[:-4]) and not k.endswith(", 0>")):
                c = v.get(counter)
                if c and c["launches"]:
                    return c["sum"] / c["launches"]  # (all launches: scaled by launches per step below)
        return None

    def traf(name):
        for k, v in traffic.items():
            if k == name or (name.endswith(">") and k.startswith(name[:-1] + ",")) or (name.endswith(", DEF>") and k.startswith(name[:-4]) and not k.endswith(", 0>")):
                return v
        return None

    rows, tot_t, tot_f = [], 0.0, 0.0
    for r in detail["roofline_kernels"]:
        name, t = r["kernel"], r["ms_per_step"]
        n = r["launches_per_step"]
        valu, mfma = pmc(name, "SQ_INSTS_VALU"), pmc(name, "SQ_INSTS_MFMA")
        floors = {}
        if valu is not None:
            # SQ_INSTS_* are per launch averaged over ALL launches of that kernel; scale to the step
            floors["issue"] = valu * ISSUE_CYCLES / (SIMDS * CLOCK) * 1e3 * n
            if mfma:
                floors["mfma"] = mfma * MFMA_CYCLES / (SIMDS * CLOCK) * 1e3 * n
        if r.get("bound") == "hbm":
            floors["hbm"] = r["bytes_per_launch"] / HBM * 1e3 * r.get("modelled_launches_per_step", n)
        elif "compulsory_hbm_GBps" in r:
            floors["hbm"] = r["compulsory_hbm_GBps"] * 1e9 * r["modelled_launch_ms"] * 1e-3 / HBM * 1e3 * r.get("modelled_launches_per_step", n)
        tv = traf(name)
        if tv and r.get("bound") in ("l2", "fabric") and tv.get("l2_misses"):
            floors["miss"] = tv["l2_misses"] / MISS_LINES * 1e3 * r.get("modelled_launches_per_step", n)
        if "hash_gathers_G_per_s" in r:
            floors["gather"] = r["hash_gathers_G_per_s"] * 1e9 * r["modelled_launch_ms"] * 1e-3 / GATHER * 1e3 * r.get("modelled_launches_per_step", n)
        if "miss" in floors and "gather" in floors:
            # round 5: the two ADD -- a line that misses L2 costs the fabric its 128 bytes on top of the gather's slot in the address path
            # (level-major kernels: time = slots / 292 G/s + misses / 64 G/s to within 10 %, profiles/r05_experiment_runs.txt)
            floors["gather+miss"] = floors["gather"] + floors["miss"]
        floor = max(floors.values()) if floors else None
        which = max(floors, key=floors.get) if floors else "-"
        rows.append((name, t, floor, which, floors, valu))
        tot_t += t
        tot_f += floor if floor is not None else t
    sep = " | " if md else "  "
    head = ["kernel", "ms/step", "floor", "roof", "gap", "VALU instr/launch", "other roofs"]
    if md:
        print("| " + " | ".join(head) + " |")
        print("|" + "---|" * len(head))
    for name, t, floor, which, floors, valu in rows:
        if t < 0.02:
            continue
        others = ", ".join("%s %.2f" % (k, v) for k, v in sorted(floors.items(), key=lambda kv: -kv[1]) if k != which)
        cells = [name[:58], "%.3f" % t, "%.2f" % floor if floor is not None else "-", which, "%.2f" % (t - floor) if floor is not None else "-",
                 "%.3g" % valu if valu else "-", others]
        print(("| " + " | ".join(cells) + " |") if md else "%-58s %8s %7s %-7s %6s %10s  %s" % tuple(cells))
    print(("| **sum** | **%.2f** | **%.2f** | | **%.2f** | | |" if md else "SUM %.2f ms  floors %.2f ms  gap %.2f ms") % (tot_t, tot_f, tot_t - tot_f))


if __name__ == "__main__":
    main()
