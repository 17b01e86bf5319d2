#!/bin/bash
# round-5 session 12: the xy stack's warped-frame requests issued with the current frame's (unconditional, shared dummy entry for reusing lanes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
python - > $O/digest.txt 2>&1 <<'PY'
import os, sys, json
sys.path.insert(0, "tests")
import test_gpu_switches as t
a = t._run({})
b = t._run({"L4D_LIB": os.path.join(os.getcwd(), "tools/abl/lib_base.so")})
print("new ", json.dumps(a)); print("prev", json.dumps(b))
print("FORWARD_IDENTICAL", all(a[k] == b[k] for k in ("depth", "image", "weights")))
PY
tail -n 3 $O/digest.txt | cut -c1-300
for v in "new X=1" "prev L4D_LIB=$PWD/tools/abl/lib_base.so"; do
  set -- $v; name=$1; shift
  env L4D_BENCH_DETAIL=$PWD/$O/${name}_detail.json "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --variant-steps 8 --trained-steps 200 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $O/${name}_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); det = json.load(open(sys.argv[2]))
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], d["config"].get("trained_state"))
for r in det["roofline_kernels"][:3]: print("   %-56s %7.3f ms" % (r["kernel"][:56], r["ms_per_step"]))
PY
done
