"""Timeline of ONE training step out of a rocprofv3 rocpd database (kernel trace): every dispatch with its queue, start offset and duration,
the time during which 0 / 1 / 2+ kernels run, and the idle gaps.  Steps are cut at `adam_ranges_kernel` (one per step).

    python tools/rocpd_timeline.py <db> [step index from the end, default 3] [min gap us to list, default 5]
"""
import sqlite3
import sys


def main(path, back=3, min_gap=5.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    q = "d.queue_id" if "queue_id" in cols else "0"
    st = "d.stream_id" if "stream_id" in cols else "0"
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, {q}, {st} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    ends = [i for i, r in enumerate(rows) if "adam_ranges_kernel" in r[0]]
    if len(ends) < back + 1:
        print("not enough steps in the trace:", len(ends))
        return
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    t_prev_end = rows[lo - 1][2]
    print(f"step of {len(step)} dispatches, {(step[-1][2] - t0) / 1e6:.3f} ms from its first start to its last end; previous step's Adam ended {(t0 - t_prev_end) / 1e3:.1f} us before")
    print(f"{'start_us':>10s} {'dur_us':>9s} {'queue':>6s} {'stream':>6s}  kernel")
    for r in step:
        print(f"{(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:9.1f} {r[3]:6d} {r[4]:6d}  {r[0][:90]}")
    # coverage: sweep over start / end events
    ev = []
    for r in step:
        ev.append((r[1], 1))
        ev.append((r[2], -1))
    ev.sort()
    busy = [0.0, 0.0, 0.0]
    depth, last = 0, ev[0][0]
    gaps = []
    for t, dlt in ev:
        if t > last:
            busy[min(depth, 2)] += t - last
            if depth == 0 and (t - last) / 1e3 >= min_gap:
                gaps.append(((last - t0) / 1e3, (t - last) / 1e3))
        depth += dlt
        last = t
    print(f"idle {busy[0] / 1e6:.3f} ms, one kernel {busy[1] / 1e6:.3f} ms, two or more {busy[2] / 1e6:.3f} ms; sum of durations {sum(r[2] - r[1] for r in step) / 1e6:.3f} ms")
    print("idle gaps >= %.0f us (offset, length): " % min_gap + ", ".join(f"{a:.0f}:{b:.1f}" for a, b in gaps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, float(sys.argv[3]) if len(sys.argv) > 3 else 5.0)
