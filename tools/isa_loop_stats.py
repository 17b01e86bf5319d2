#!/usr/bin/env python3
"""Instruction mix of a kernel's loops from hipcc's assembly (``hipcc --cuda-device-only -S``): for every backward branch
target (= loop head) the instructions between the label and the branch are counted by class -- what a kernel bound by issue
slots spends them on, without a GPU.

    python tools/isa_loop_stats.py file.s '<kernel symbol substring>' [min_instructions]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_cvt"):
        return "v_cvt"
    if op.startswith("v_pk_"):
        return "v_pk"
    if "dpp" in op:
        return "v_dpp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem_" + ("atomic" if "atomic" in op else "store" if "store" in op else "load")
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.rstrip().split(":")[0].find(key) >= 0 and ":" in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels, insts = {}, []
    for l in body:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        if "dpp" in s or "row_shr" in s or "row_bcast" in s or "quad_perm" in s:
            op += "_dpp"
        insts.append((op, s))
    print("%s: %d instructions" % (lines[start].split(":")[0][:100], len(insts)))
    for idx, (op, s) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= idx and idx - labels[tgt] >= min_n:
                seg = insts[labels[tgt]:idx + 1]
                c = Counter(classify(o) for o, _ in seg)
                print("  loop %s: %d instructions  " % (tgt, len(seg)) + "  ".join("%s=%d" % kv for kv in c.most_common()))
                if "-v" in sys.argv:
                    ops = Counter(o for o, _ in seg)
                    print("     " + "  ".join("%s=%d" % kv for kv in ops.most_common(40)))


if __name__ == "__main__":
    main()
