#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in hipcc's device assembly (``hipcc --cuda-device-only -S file.hip -o file.s``),
without a GPU:   python tools/kernel_regs.py file.s [name substring]"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, body = m.group(1), m.group(2)
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        if want not in name:
            continue
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        acc = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        print("%-100s vgpr+agpr %3d (arch %s) sgpr %3d lds %6d scratch %d" % (name.split("(")[0][:100], get("next_free_vgpr"), acc.group(1) if acc else "-",
                                                                          get("next_free_sgpr"), get("group_segment_fixed_size"), get("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
