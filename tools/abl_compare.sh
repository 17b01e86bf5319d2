#!/bin/bash
# usage: tools/abl_compare.sh lib1.so lib2.so ...  -- runs the default bench with each library build (L4D_LIB) and prints
# step time + the per-entry-point kernel times of interest.  "default" = the in-tree library.
for lib in "$@"; do
  if [ "$lib" = default ]; then unset L4D_LIB; else export L4D_LIB=$PWD/$lib; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_per_step']
print('$lib', 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s=%.2f' % (n.replace('l4d_',''), k[n]) for n in list(k)[:6]))"
done
