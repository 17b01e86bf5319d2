"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                       f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr agpr sgpr lds scratch")
    for r in rows[:top]:
        name = r[0][:70]
        print(f"{name:70s} {r[1]:7d} {r[2]/1e6:10.3f} {r[2]/r[1]/1e3:10.1f} {r[3]/1e3:9.1f} {r[4]/1e3:9.1f} {100*r[2]/total:6.2f} {r[5]} {r[6]} {r[7]} {r[8]} {r[9]}")
    print(f"total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
