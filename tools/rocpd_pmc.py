"""Per-kernel sums/averages of the PMC counters in a rocprofv3 rocpd SQLite database.

    python tools/rocpd_pmc.py results.db [--json out.json]

--json: {kernel (demangled, as written at the launch site, e.g. "mlp_fwd_kernel<8, 1>"): {counter: {"launches": n, "sum": s,
         "big_launches": nb, "big_sum": sb}}} -- big_*: only the dispatches that took at least half as long as the kernel's longest one
         (a step launches some kernels on the render batch AND on small inputs -- the scene-flow loss evaluates the flow field on
         a frame's point cloud --; a byte model of the render-sized launch must be compared with those launches only)
"""
import json
import re
import shutil
import sqlite3
import subprocess
import sys


def simple_demangle(n):
    """Itanium name of a (possibly truncated) kernel symbol -> "name<args>" for integral / bool template arguments, which is
    all this library uses; anything else is returned unchanged.  (rocprofv3 stores long symbols truncated, which c++filt
    rejects.)"""
    m = re.match(r"_Z(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    name = n[m.end():m.end() + ln]
    rest = n[m.end() + ln:]
    if not rest.startswith("I"):
        return name
    args, i = [], 1
    while i < len(rest) and rest[i] == "L":
        mm = re.match(r"L([a-z])(n?\d+)E", rest[i:])
        if not mm:
            return n
        t, v = mm.group(1), mm.group(2).replace("n", "-")
        args.append(("true" if v != "0" else "false") if t == "b" else v)
        i += mm.end()
    return f"{name}<{', '.join(args)}>" if i < len(rest) and rest[i] == "E" else n


def demangle(names):
    return {n: simple_demangle(n.replace(".kd", "")) for n in names}


def demangle_cxxfilt(names):
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([filt], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True, check=True).stdout.splitlines()
    except Exception:
        return {n: n for n in names}
    res = {}
    for n, d in zip(names, out):
        d = re.sub(r"^void\s+", "", d.strip())
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):  # cut the argument list: the first "(" outside template brackets
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        res[n] = d[:cut]
    return res


def main(path, json_out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols_ip = [r[1] for r in cur.execute(f"pragma table_info({ip})")]
    name_col = "name" if "name" in cols_ip else cols_ip[1]
    q = (f"select s.kernel_name, p.{name_col}, count(distinct d.id), sum(e.value), sum(d.end - d.start) / count(distinct e.pmc_id) "
         f"from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
         f"group by s.kernel_name, p.{name_col} order by 4 desc")
    rows = cur.execute(q).fetchall()
    big = {}
    cols_kd = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    # "large" launches of a kernel: at least half as long as its longest one (grids do not tell: several kernels cap their grid)
    if "start" in cols_kd and "end" in cols_kd:
        qb = (f"select s.kernel_name, p.{name_col}, count(distinct d.id), sum(e.value) "
              f"from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
              f"join (select kernel_id as kid, max(end - start) as md from {kd} group by kernel_id) g on g.kid = d.kernel_id "
              f"where (d.end - d.start) * 2 >= g.md group by s.kernel_name, p.{name_col}")
        for kname, cname, n, ssum in cur.execute(qb).fetchall():
            big[(kname, str(cname))] = (n, ssum)
    names = demangle(sorted({r[0] for r in rows}))
    print(f"{'kernel':56s} {'counter':28s} {'launches':>8s} {'sum':>14s} {'per_launch':>14s} {'kernel_ns/launch':>16s}")
    for r in rows[:240]:
        print(f"{names[r[0]][:56]:56s} {str(r[1]):28s} {r[2]:8d} {r[3]:14.4e} {r[3] / max(r[2], 1):14.4e} {r[4] / max(r[2], 1):16.0f}")
    if json_out:
        out = {}
        for r in rows:
            e = {"launches": r[2], "sum": r[3], "kernel_ns": r[4]}
            if (r[0], str(r[1])) in big:
                e["big_launches"], e["big_sum"] = big[(r[0], str(r[1]))]
            out.setdefault(names[r[0]], {})[str(r[1])] = e
        json.dump(out, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None)
