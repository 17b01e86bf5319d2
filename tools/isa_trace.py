#!/usr/bin/env python3
"""Memory-operation skeleton of one kernel in hipcc's device assembly: loads, stores, atomics, s_waitcnt, barriers and loop heads in
program order -- enough to see, without a GPU, whether a kernel's loads are in flight together or waited for one by one.

    python tools/isa_trace.py file.s '<kernel symbol substring>' [--lds]
"""
import re
import sys


def main():
    text = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    lds = "--lds" in sys.argv
    start = next(i for i, l in enumerate(text) if re.match(r"^[A-Za-z_][\w$.]*:", l) and want in l and not l.startswith(".L"))
    n = 0
    for l in text[start + 1:]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        n += 1
        keep = t.startswith(("global_", "buffer_", "flat_", "scratch_", "s_barrier", "s_endpgm")) or "vmcnt" in t or "Loop Header" in t
        if lds and (t.startswith("ds_") or "lgkmcnt" in t):
            keep = True
        if keep:
            print(n, t[:110])
        if t.startswith("s_endpgm"):
            break


if __name__ == "__main__":
    main()
