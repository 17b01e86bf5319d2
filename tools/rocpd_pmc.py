"""Per-kernel sums/averages of the PMC counters in a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols_pe = [r[1] for r in cur.execute(f"pragma table_info({pe})")]
    cols_ip = [r[1] for r in cur.execute(f"pragma table_info({ip})")]
    print("# pmc_event columns:", cols_pe)
    print("# info_pmc columns:", cols_ip)
    name_col = "name" if "name" in cols_ip else cols_ip[1]
    q = (f"select s.kernel_name, p.{name_col}, count(distinct d.id), sum(e.value), sum(d.end - d.start) / count(distinct e.pmc_id) "
         f"from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
         f"group by s.kernel_name, p.{name_col} order by 4 desc")
    print(f"{'kernel':64s} {'counter':12s} {'launches':>8s} {'sum':>14s} {'per_launch':>14s}")
    for r in cur.execute(q).fetchall()[:200]:
        print(f"{r[0][:64]:64s} {str(r[1]):12s} {r[2]:8d} {r[3]:14.4e} {r[3] / max(r[2], 1):14.4e}")


if __name__ == "__main__":
    main(sys.argv[1])
