#!/usr/bin/env python3
"""Opcode histogram of a kernel's LARGEST loop (hipcc --cuda-device-only -S):  python tools/isa_hist.py file.s '<mangled name prefix>' [top]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
i = s.index("\n" + sys.argv[2])
j = s.index("s_endpgm", i)
body = s[i:j].splitlines()
labels = {}
for k, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = k
best = (0, 0, 0)
for k, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > best[0]:
        best = (k - labels[m.group(1)], labels[m.group(1)], k)
ops = collections.Counter()
for l in body[best[1]:best[2]]:
    l = l.strip()
    if l and not l.startswith((".", ";")):
        ops[l.split()[0]] += 1
print("largest loop: %d instructions" % sum(ops.values()))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print(f"  {k:28s} {v}")
