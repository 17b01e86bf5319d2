#!/usr/bin/env python3
"""Bottom-up floor of the training step: per kernel, the time its BINDING roof would allow, next to the time it takes.

    python tools/floor_table.py <bench_detail.json> <pmc_inst.json> [<hbm_traffic.json>] [--stats <kernel_stats.txt>] [--md]

Roofs (all measured on this chip):
  issue   every wave-level VALU instruction needs issue time on its SIMD, and how much depends on how many wavefronts share the SIMD
          (tools/ubench/valu_rate.hip, profiles/r06_ubench_valu.txt): ONE wavefront issues an instruction every 4.75 cycles
          whatever it is; with w wavefronts the SIMD spends, per wave-instruction, 2.38 / 1.68 / 1.33 cycles at w = 2 / 4 / 8 for the
          plain ops (v_fma_f32, v_add_u32, v_xor_b32, compare + select pairs) and 3.5 / 2.75 / 2.0 for conversions, DPP, v_mul_lo,
          3-operand and packed ops, shifts of 64 bits.  The FLOOR uses the plain-op rate at the kernel's occupancy (a true lower
          bound); `issue~` in the last column is the same count at the other rate (what a DPP- / conversion-heavy kernel really pays).
          Rounds 4-5 charged 4 cycles flat (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.00 quad-cycles: a counter that cannot read
          lower) -- right at two wavefronts per SIMD, 2-3x too high at eight.  Occupancy: `w/simd` of the kernel-trace summary
          (--stats, tools/rocpd_stats.py), else the table below.
          MFMA kernels: + SQ_INSTS_MFMA x 16 cycles (v_mfma_f32_16x16x32_f16) on the matrix pipe, beside the VALU -> max of the two.
  hbm     compulsory bytes of the byte model (bench.py kernel_models) / 6.3 TB/s achievable.
  miss    gather kernels: lines that miss an XCD's L2 arrive at 64 G lines/s (TCC_MISS of the traffic file).
  gather  gather kernels: 292 G distinct-address slots/s when every line hits (table-entry gathers only).
  lds     LDS integer atomics: lane-operations of the kernel's accumulation scheme (per sample, from the kernels' own design:
          the table below) x samples / the measured rate on random addresses (profiles/r01_ubench_lds_atomics.txt: ds_add_u32
          4.8 T lane-ops/s, ds_add_u64 2.9 T/s chip-wide).  An upper bound of the count where zero contributions are skipped.
  taps    the fused encode's coherent plane taps: 240 vector loads per wavefront and 64 samples, >= 18 clocks of the CU's address
          path each however few texels they touch (tools/ubench/gather.hip run-length rows).
  gather+miss(+taps)  the sum (round 5: they add: level-major kernels run at slots / 292 G/s + misses / 64 G/s to within 10 %).
floor = max of the applicable ones, CLAMPED to the measured time (a model that a kernel beats is not a floor: round 5's table had
two negative gaps); gap = time - floor.  The sum of the floors is what THIS design (this instruction stream, this traffic) could
reach if every kernel sat on its roof -- not a lower bound for the problem.
"""
import json
import re
import sys

CLOCK = 2.3e9          # sustained shader clock under these kernels (GRBM_GUI_ACTIVE / wall, profiles/*_pmc_mfma)
SIMDS = 1024
CUS = 256
# SIMD cycles per wave-level VALU instruction by wavefronts per SIMD (profiles/r06_ubench_valu.txt)
ISSUE_PLAIN = {1: 4.75, 2: 2.38, 3: 1.95, 4: 1.68, 5: 1.56, 6: 1.46, 7: 1.39, 8: 1.33}
ISSUE_OTHER = {1: 5.0, 2: 3.5, 3: 3.05, 4: 2.75, 5: 2.5, 6: 2.3, 7: 2.13, 8: 2.0}
MFMA_CYCLES = 16.0     # v_mfma_f32_16x16x32_f16 on one SIMD
HBM = 6.3e12
MISS_LINES = 64e9
GATHER = 292e9
TAP_LOADS_PER_WAVE, TAP_CLOCKS = 240, 18.0
# LDS atomic lane-operations per SAMPLE and the rate that applies (the accumulation scheme of each kernel)
LDS_U32, LDS_U64 = 4.8e12, 2.9e12
LDS_ATOMICS = {"bin_reduce_kernel<3, 4": (8 * 4 * 2 * 4, LDS_U64),   # 8 levels x 4 pair records x 2 entries x 4 values
               "bin_reduce_kernel<3, 2": (7 * 4 * 2 * 2, LDS_U64),   # flow grid: 7 binned levels, 2 values (upper bound: merged runs emit fewer)
               "planes_static_lds_kernel": (384, LDS_U32),            # 3 planes x 4 scales x 4 texels x 8 channels
               "dynhash_lds_kernel": (3 * 8 * 4, LDS_U64),            # 3 planes x 8 levels x 4 corners, one scalar per entry
               "planes_dyn_lds_kernel": (50, LDS_U32),                # after the segmented row scans (measured average)
               "bin_pass1_kernel<3, 4": (8 * 4, LDS_U32), "bin_pass1_kernel<3, 2": (7 * 4, LDS_U32)}  # ranks (returning)
# wavefronts per SIMD where no kernel-trace summary is given (registers / LDS / workgroup size of the default build)
OCCUPANCY = {"density_encode_fwd_kernel": 2, "planes_dyn_lds_kernel": 3, "mlp_bwd_kernel<6": 1, "mlp_bwd_kernel<8": 1, "mlp_bwd_kernel<1": 2,
             "dynhash_fwd_lds_kernel": 4, "bin_pass1_kernel": 4, "bin_reduce_kernel": 8, "hashgrid_fwd_levels_kernel": 8,
             "planes_static_lds_kernel": 8, "hashgrid_t_fwd_levels_kernel": 6, "dynhash_lds_kernel": 4, "mlp_fwd_kernel": 4}


def occupancy_from_stats(path):
    """kernel (mangled prefix) -> wavefronts per SIMD from tools/rocpd_stats.py's last column"""
    occ = {}
    for line in open(path):
        f = line.split()
        if len(f) > 13 and f[0].startswith("_Z") and re.match(r"^[0-9.]+$", f[-1]):
            m = re.match(r"_Z(\d+)", f[0])
            if m:
                occ[f[0][m.end():m.end() + int(m.group(1))]] = max(1, min(8, int(round(float(f[-1])))))
    return occ


def main():
    argv = sys.argv[1:]
    stats_occ = {}
    if "--stats" in argv:
        k = argv.index("--stats")
        stats_occ = occupancy_from_stats(argv[k + 1])
        del argv[k:k + 2]
    args = [a for a in argv if not a.startswith("--")]
    detail = json.load(open(args[0]))

    def waves(name):
        base = name.split("<")[0]
        if base in stats_occ and not name.startswith("mlp_"):  # (the MLP kernels share a base name across shapes: table below)
            return stats_occ[base]
        for k, v in sorted(OCCUPANCY.items(), key=lambda kv: -len(kv[0])):
            if name.startswith(k):
                return v
        return stats_occ.get(base, 4)

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
        w = waves(name)
        issue_other = None
        if valu is not None:
            # SQ_INSTS_* are per launch averaged over ALL launches of that kernel; scale to the step
            floors["issue"] = valu * ISSUE_PLAIN[w] / (SIMDS * CLOCK) * 1e3 * n
            issue_other = valu * ISSUE_OTHER[w] / (SIMDS * CLOCK) * 1e3 * n
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
        for k, (per_sample, rate) in LDS_ATOMICS.items():
            samples = detail.get("roofline", {}).get("samples_per_launch") or 0
            if name.startswith(k) and samples:
                floors["lds"] = per_sample * samples / rate * 1e3
        if name.startswith("density_encode_fwd_kernel") and r.get("bytes_per_launch"):
            samples = detail.get("roofline", {}).get("samples_per_launch") or 0
            if samples:
                floors["taps"] = samples / 64.0 * TAP_LOADS_PER_WAVE * TAP_CLOCKS / (CUS * CLOCK) * 1e3 * r.get("modelled_launches_per_step", n)
        if "miss" in floors and "gather" in floors:
            # round 5: the two ADD -- a line that misses L2 costs the fabric its 128 bytes on top of the gather's slot in the address path
            # (level-major kernels: time = slots / 292 G/s + misses / 64 G/s to within 10 %, profiles/r05_experiment_runs.txt)
            key = "gather+miss+taps" if "taps" in floors else "gather+miss"
            floors[key] = floors["gather"] + floors["miss"] + floors.get("taps", 0.0)
        floor = max(floors.values()) if floors else None
        which = max(floors, key=floors.get) if floors else "-"
        if floor is not None and floor > t:  # a model the kernel beats is not a floor
            floor, which = t, which + " (clamped)"
        if issue_other is not None:
            floors["issue~"] = issue_other
        floors["w/simd"] = w
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
        others = "w/simd %d, " % floors.pop("w/simd") + ", ".join("%s %.2f" % (k, v) for k, v in sorted(floors.items(), key=lambda kv: -kv[1]) if k != which.replace(" (clamped)", ""))
        cells = [name[:58], "%.3f" % t, "%.2f" % floor if floor is not None else "-", which, "%.2f" % (t - floor) if floor is not None else "-",
                 "%.3g" % valu if valu else "-", others]
        print(("| " + " | ".join(cells) + " |") if md else "%-58s %8s %7s %-7s %6s %10s  %s" % tuple(cells))
    print(("| **sum** | **%.2f** | **%.2f** | | **%.2f** | | |" if md else "SUM %.2f ms  floors %.2f ms  gap %.2f ms") % (tot_t, tot_f, tot_t - tot_f))


if __name__ == "__main__":
    main()
