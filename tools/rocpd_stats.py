"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = {r[1] for r in cur.execute(f"pragma table_info({kd})")}
    wg = "max(d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z)" if {"workgroup_size_x", "workgroup_size_y", "workgroup_size_z"} <= cols else "0"
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                       f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size), {wg} "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    # wg = threads per workgroup; w/simd = wavefronts per SIMD the kernel can hold: min(8, registers, LDS, 32 waves per CU) -- the vgpr
    # column of rocprofv3 counts in units of two registers on gfx950 (allocation granule 8: MI355X_MICROARCH.md)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr agpr sgpr lds scratch wg w/simd")
    for r in rows[:top]:
        name = r[0][:70]
        regs = -(-(2 * (r[5] + r[6])) // 8) * 8 if (r[5] + r[6]) else 8
        occ = "-"
        if r[10]:
            waves_wg = -(-r[10] // 64)
            wgs = min(32 // waves_wg if waves_wg else 0, (160 * 1024) // r[8] if r[8] else 99, (4 * min(8, 512 // regs)) // waves_wg if waves_wg <= 4 * min(8, 512 // regs) else 0)
            occ = "%.1f" % (wgs * waves_wg / 4.0)
        print(f"{name:70s} {r[1]:7d} {r[2]/1e6:10.3f} {r[2]/r[1]/1e3:10.1f} {r[3]/1e3:9.1f} {r[4]/1e3:9.1f} {100*r[2]/total:6.2f} {r[5]} {r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {occ}")
    print(f"total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
